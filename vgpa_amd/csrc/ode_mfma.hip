// Dispatcher of the symmetric fp64-MFMA stepping kernels (kernels: ode_mfma_impl.h, one TU per stepper).
#include "ode_mfma_impl.h"

namespace vgpa {

bool ode_mfma_supported(int method, bool, int D) {
  if (D < 1) return false;
  const int nb = (D + 3) / 4;
  if (nb > 16) return false;   // sym::kMaxNB: D <= 64
  switch (method) {
    case VGPA_ODE_EULER: return mfma_method_supported<VGPA_ODE_EULER>(nb);
    case VGPA_ODE_HEUN: return mfma_method_supported<VGPA_ODE_HEUN>(nb);
    case VGPA_ODE_RK2: return mfma_method_supported<VGPA_ODE_RK2>(nb);
    case VGPA_ODE_RK4: return mfma_method_supported<VGPA_ODE_RK4>(nb);
  }
  return false;
}

bool sym_stores_q(int method, int D) { return sym::stores_q(method, D); }
bool sym_fuses_grad(int method, int D) { return sym::fuses_grad(method, D); }

hipError_t launch_ode_mfma(int method, bool fwd, const OdeArgs& a, hipStream_t st) {
  switch (method) {
    case VGPA_ODE_EULER: return mfma_method_launch<VGPA_ODE_EULER>(fwd, a, st);
    case VGPA_ODE_HEUN: return mfma_method_launch<VGPA_ODE_HEUN>(fwd, a, st);
    case VGPA_ODE_RK2: return mfma_method_launch<VGPA_ODE_RK2>(fwd, a, st);
    case VGPA_ODE_RK4: return mfma_method_launch<VGPA_ODE_RK4>(fwd, a, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace vgpa
