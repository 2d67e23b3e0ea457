// Optional library backend for the PLAIN fp64 GEMM of a large-D RK stage (W = A.X, W' = A^T.Psi, one rank owning all
// rows): rocBLAS dgemm, loaded with dlopen on first use so that libvgpa_hip.so itself has no link-time dependency.
// Selected by VGPA_FLAG_LIBRARY_GEMM; the default is the hand-written MFMA kernel of large_d.hip (k_gemm), which this
// also serves as a yardstick for.  Everything fused (stage kernel, mid-point operand, packed
// row-sharded output) stays hand-written.
#include <dlfcn.h>

#include <map>
#include <mutex>

#include "vgpa_internal.h"

namespace vgpa {
namespace ld {
namespace {

typedef int (*create_handle_t)(void**);
typedef int (*set_stream_t)(void*, hipStream_t);
typedef int (*dgemm_t)(void*, int, int, int, int, int, const double*, const double*, int, const double*, int, const double*,
                       double*, int);
constexpr int kOpNone = 111, kOpTrans = 112;      // rocblas_operation_none / rocblas_operation_transpose

struct Api {
  void* so = nullptr;
  create_handle_t create = nullptr;
  std::map<int, void*> handles;      // one rocBLAS handle per device (a handle is bound to the device current at creation)
  set_stream_t set_stream = nullptr;
  dgemm_t dgemm = nullptr;
  bool tried = false;
};
Api g_api;
std::mutex g_mu;

bool load_api() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_api.tried) return g_api.create != nullptr;
  g_api.tried = true;
  const char* names[] = {"librocblas.so", "librocblas.so.5", "librocblas.so.4", "/opt/rocm/lib/librocblas.so"};
  for (const char* n : names) {
    g_api.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // reuse a copy the process already holds (e.g. torch's)
    if (g_api.so) break;
  }
  for (const char* n : names) {
    if (g_api.so) break;
    g_api.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!g_api.so) return false;
  auto create = (create_handle_t)dlsym(g_api.so, "rocblas_create_handle");
  g_api.set_stream = (set_stream_t)dlsym(g_api.so, "rocblas_set_stream");
  g_api.dgemm = (dgemm_t)dlsym(g_api.so, "rocblas_dgemm");
  if (!create || !g_api.set_stream || !g_api.dgemm) return false;
  g_api.create = create;
  return true;
}

// the handle of the device that is current now (the entry points select the context's device before they get here)
void* device_handle() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_api.handles.find(dev);
  if (it != g_api.handles.end()) return it->second;
  void* h = nullptr;
  if (g_api.create(&h) != 0) h = nullptr;
  g_api.handles[dev] = h;
  return h;
}

}  // namespace

bool library_gemm_available() { return load_api(); }

// Row-major C[M x N] = op(A) . B,  op(A) = A ([M x K], lda) or A^T with A stored [K x M] (lda); B [K x N] (ldb).
// Column-major view: C^T = B^T . op(A)^T, and a row-major matrix IS the column-major transpose.
hipError_t library_gemm(bool transa, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                        hipStream_t st) {
  if (!load_api()) return hipErrorNotSupported;
  void* handle = device_handle();
  if (!handle) return hipErrorNotSupported;
  if (g_api.set_stream(handle, st) != 0) return hipErrorUnknown;
  const double one = 1.0, zero = 0.0;
  const int rc = g_api.dgemm(handle, kOpNone, transa ? kOpTrans : kOpNone, N, M, K, &one, B, ldb, A, lda, &zero, C, ldc);
  return rc == 0 ? hipSuccess : hipErrorUnknown;
}

}  // namespace ld
}  // namespace vgpa
