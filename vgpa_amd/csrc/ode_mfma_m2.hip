// Instantiations of the symmetric fp64-MFMA stepping kernels for stepper 2 (see ode_mfma_impl.h).
#include "ode_mfma_impl.h"
namespace vgpa {
template <> bool mfma_method_supported<2>(int nb) {
  switch (nb) {
    case 1:
    case 3:
    case 5:
    case 10:
      return true;
    default: return false;
  }
}
template <> hipError_t mfma_method_launch<2>(bool fwd, const OdeArgs& a, hipStream_t st) {
  switch ((a.D + 3) / 4) {
    case 1: return fwd ? mfma::launch_nb<2, true, 1>(a, st) : mfma::launch_nb<2, false, 1>(a, st);
    case 3: return fwd ? mfma::launch_nb<2, true, 3>(a, st) : mfma::launch_nb<2, false, 3>(a, st);
    case 5: return fwd ? mfma::launch_nb<2, true, 5>(a, st) : mfma::launch_nb<2, false, 5>(a, st);
    case 10: return fwd ? mfma::launch_nb<2, true, 10>(a, st) : mfma::launch_nb<2, false, 10>(a, st);
    default: return hipErrorNotSupported;
  }
}
}  // namespace vgpa
