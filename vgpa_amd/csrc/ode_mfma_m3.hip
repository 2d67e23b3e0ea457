// Instantiations of the symmetric fp64-MFMA stepping kernels for stepper 3 (see ode_mfma_impl.h).
#include "ode_mfma_impl.h"
namespace vgpa {
template <> bool mfma_method_supported<3>(int nb) {
  switch (nb) {
    case 1:
    case 2:
    case 3:
    case 4:
    case 5:
    case 6:
    case 7:
    case 8:
    case 9:
    case 10:
    case 11:
      return true;
    default: return false;
  }
}
template <> hipError_t mfma_method_launch<3>(bool fwd, const OdeArgs& a, hipStream_t st) {
  switch ((a.D + 3) / 4) {
    case 1: return fwd ? mfma::launch_nb<3, true, 1>(a, st) : mfma::launch_nb<3, false, 1>(a, st);
    case 2: return fwd ? mfma::launch_nb<3, true, 2>(a, st) : mfma::launch_nb<3, false, 2>(a, st);
    case 3: return fwd ? mfma::launch_nb<3, true, 3>(a, st) : mfma::launch_nb<3, false, 3>(a, st);
    case 4: return fwd ? mfma::launch_nb<3, true, 4>(a, st) : mfma::launch_nb<3, false, 4>(a, st);
    case 5: return fwd ? mfma::launch_nb<3, true, 5>(a, st) : mfma::launch_nb<3, false, 5>(a, st);
    case 6: return fwd ? mfma::launch_nb<3, true, 6>(a, st) : mfma::launch_nb<3, false, 6>(a, st);
    case 7: return fwd ? mfma::launch_nb<3, true, 7>(a, st) : mfma::launch_nb<3, false, 7>(a, st);
    case 8: return fwd ? mfma::launch_nb<3, true, 8>(a, st) : mfma::launch_nb<3, false, 8>(a, st);
    case 9: return fwd ? mfma::launch_nb<3, true, 9>(a, st) : mfma::launch_nb<3, false, 9>(a, st);
    case 10: return fwd ? mfma::launch_nb<3, true, 10>(a, st) : mfma::launch_nb<3, false, 10>(a, st);
    case 11: return fwd ? mfma::launch_nb<3, true, 11>(a, st) : mfma::launch_nb<3, false, 11>(a, st);
    default: return hipErrorNotSupported;
  }
}
}  // namespace vgpa
