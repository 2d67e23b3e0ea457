// Instantiations of the symmetric fp64-MFMA stepping kernels for stepper 1: D <= 44 on the role-specialised kernels of
// ode_mfma_impl.h (VGPA_ODE_KERNEL=sym selects the symmetric-unit ones), 44 < D <= 64 on the symmetric-unit kernels of
// ode_sym_impl.h.
#include "ode_sym_impl.h"
namespace vgpa {
template <> bool mfma_method_supported<1>(int nb) { return nb >= 1 && nb <= sym::kMaxNB; }
template <> hipError_t mfma_method_launch<1>(bool fwd, const OdeArgs& a, hipStream_t st) {
  switch ((a.D + 3) / 4) {
    case 1: return fwd ? sym::launch_any<1, true, 1>(a, st) : sym::launch_any<1, false, 1>(a, st);
    case 2: return fwd ? sym::launch_any<1, true, 2>(a, st) : sym::launch_any<1, false, 2>(a, st);
    case 3: return fwd ? sym::launch_any<1, true, 3>(a, st) : sym::launch_any<1, false, 3>(a, st);
    case 4: return fwd ? sym::launch_any<1, true, 4>(a, st) : sym::launch_any<1, false, 4>(a, st);
    case 5: return fwd ? sym::launch_any<1, true, 5>(a, st) : sym::launch_any<1, false, 5>(a, st);
    case 6: return fwd ? sym::launch_any<1, true, 6>(a, st) : sym::launch_any<1, false, 6>(a, st);
    case 7: return fwd ? sym::launch_any<1, true, 7>(a, st) : sym::launch_any<1, false, 7>(a, st);
    case 8: return fwd ? sym::launch_any<1, true, 8>(a, st) : sym::launch_any<1, false, 8>(a, st);
    case 9: return fwd ? sym::launch_any<1, true, 9>(a, st) : sym::launch_any<1, false, 9>(a, st);
    case 10: return fwd ? sym::launch_any<1, true, 10>(a, st) : sym::launch_any<1, false, 10>(a, st);
    case 11: return fwd ? sym::launch_any<1, true, 11>(a, st) : sym::launch_any<1, false, 11>(a, st);
    case 12: return fwd ? sym::launch_sym<1, true, 12>(a, st) : sym::launch_sym<1, false, 12>(a, st);
    case 13: return fwd ? sym::launch_sym<1, true, 13>(a, st) : sym::launch_sym<1, false, 13>(a, st);
    case 14: return fwd ? sym::launch_sym<1, true, 14>(a, st) : sym::launch_sym<1, false, 14>(a, st);
    case 15: return fwd ? sym::launch_sym<1, true, 15>(a, st) : sym::launch_sym<1, false, 15>(a, st);
    case 16: return fwd ? sym::launch_sym<1, true, 16>(a, st) : sym::launch_sym<1, false, 16>(a, st);
    default: return hipErrorNotSupported;
  }
}
}  // namespace vgpa
