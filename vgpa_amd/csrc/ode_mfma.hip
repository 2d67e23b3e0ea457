// Symmetric fast path of the time-stepping kernels (fp64 MFMA).  Placeholder: not built yet.
#include "vgpa_internal.h"
namespace vgpa {
bool ode_mfma_supported(int, bool, int) { return false; }
hipError_t launch_ode_mfma(int, bool, const OdeArgs&, hipStream_t) { return hipErrorNotSupported; }
}  // namespace vgpa
