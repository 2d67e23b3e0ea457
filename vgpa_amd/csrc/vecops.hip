// Device-resident vector algebra for the SCG driver: the optimiser's x, d, gradients never
// leave HBM; only scalars (dot products, max-abs) come back to the host that runs the unchanged SCG control flow
// (src/numerics/optim_scg.py:75-285).  Every operation is segmented: `nseg` independent problems of `seglen` doubles
// each (the batch of a context), one scalar per segment.  Reductions are deterministic: fixed-shape two-level trees.
#include "vgpa_internal.h"

namespace vgpa {
namespace {

constexpr int NT = 256;

__device__ __forceinline__ double combine(int mode, double x, double y) { return (mode == 1) ? fmax(x, y) : x + y; }

// mode 0: sum a*b   1: max |a|   2: sum |a|.  grid = (blocks per segment, nseg); part[seg][gridDim.x]
__global__ void __launch_bounds__(NT) k_reduce1(int mode, const double* a, const double* b, long long seglen, double* part) {
  __shared__ double red[NT];
  const long long base = (long long)blockIdx.y * seglen;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < seglen; i += (long long)gridDim.x * NT) {
    const double v = a[base + i];
    if (mode == 0) acc = __builtin_fma(v, b[base + i], acc);
    else if (mode == 1) acc = fmax(acc, fabs(v));
    else acc += fabs(v);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = combine(mode, red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(NT) k_reduce2(int mode, const double* part, int np, double* out) {
  __shared__ double red[NT];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += NT) acc = combine(mode, acc, part[(size_t)blockIdx.x * np + i]);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = combine(mode, red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// out = alpha[seg] * x + beta[seg] * y   (y may be nullptr: treated as 0; out may alias x or y)
__global__ void __launch_bounds__(NT) k_axpby(long long seglen, const double* alpha, const double* x, const double* beta,
                                              const double* y, double* out) {
  const long long base = (long long)blockIdx.y * seglen;
  const double al = alpha[blockIdx.y], be = y ? beta[blockIdx.y] : 0.0;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < seglen; i += (long long)gridDim.x * NT) {
    const double xv = x[base + i];
    // a zero coefficient must drop its operand even if that operand holds inf/nan
    double r = (al == 0.0) ? 0.0 : al * xv;
    if (be != 0.0) r += be * y[base + i];
    out[base + i] = r;
  }
}

}  // namespace

int vec_blocks_per_seg(long long seglen, int nseg) {
  long long nb = (seglen + NT - 1) / NT;
  long long cap = 2048 / (nseg < 1 ? 1 : nseg);
  if (cap < 1) cap = 1;
  if (cap > 256) cap = 256;
  return (int)(nb < cap ? nb : cap);
}

// scratch: [nseg results][nseg * blocks partials]
hipError_t vec_reduce(int mode, const double* a, const double* b, int nseg, long long seglen, double* scratch, hipStream_t st) {
  const int bps = vec_blocks_per_seg(seglen, nseg);
  hipLaunchKernelGGL(k_reduce1, dim3(bps, nseg), dim3(NT), 0, st, mode, a, b, seglen, scratch + nseg);
  hipLaunchKernelGGL(k_reduce2, dim3(nseg), dim3(NT), 0, st, mode, scratch + nseg, bps, scratch);
  return hipGetLastError();
}

hipError_t vec_axpby(int nseg, long long seglen, const double* alpha_dev, const double* x, const double* beta_dev,
                     const double* y, double* out, hipStream_t st) {
  long long nb = (seglen + NT - 1) / NT;
  long long cap = 8192 / nseg; if (cap < 1) cap = 1;
  const int bps = (int)(nb < cap ? nb : cap);
  hipLaunchKernelGGL(k_axpby, dim3(bps, nseg), dim3(NT), 0, st, seglen, alpha_dev, x, beta_dev, y, out);
  return hipGetLastError();
}

}  // namespace vgpa
