// Optional library backend for the PLAIN fp64 GEMM of a large-D RK stage (W = A.X, W' = A^T.Psi, one rank owning all
// rows): rocBLAS dgemm, loaded with dlopen on first use so that libvgpa_hip.so itself has no link-time dependency.
// Selected by VGPA_FLAG_LIBRARY_GEMM; the default is the hand-written MFMA kernel of large_d.hip (k_gemm), which this
// also serves as a yardstick for.  Everything fused (stage kernel, mid-point operand, packed
// row-sharded output) stays hand-written.
#include <dlfcn.h>

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
  void* handle = nullptr;
  set_stream_t set_stream = nullptr;
  dgemm_t dgemm = nullptr;
  bool tried = false;
};
Api g_api;
std::mutex g_mu;

bool load_api() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_api.tried) return g_api.handle != nullptr;
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
  if (create(&g_api.handle) != 0) { g_api.handle = nullptr; return false; }
  return true;
}

}  // namespace

bool library_gemm_available() { return load_api(); }

// Row-major C[M x N] = op(A) . B,  op(A) = A ([M x K], lda) or A^T with A stored [K x M] (lda); B [K x N] (ldb).
// Column-major view: C^T = B^T . op(A)^T, and a row-major matrix IS the column-major transpose.
hipError_t library_gemm(bool transa, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                        hipStream_t st) {
  if (!load_api()) return hipErrorNotSupported;
  if (g_api.set_stream(g_api.handle, st) != 0) return hipErrorUnknown;
  const double one = 1.0, zero = 0.0;
  const int rc = g_api.dgemm(g_api.handle, kOpNone, transa ? kOpTrans : kOpNone, N, M, K, &one, B, ldb, A, lda, &zero, C, ldc);
  return rc == 0 ? hipSuccess : hipErrorUnknown;
}

}  // namespace ld
}  // namespace vgpa
