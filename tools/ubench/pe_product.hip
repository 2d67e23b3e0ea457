// Micro-benchmark of the P-wave product alone: cycles per product for a lone wave per SIMD, with / without the phase barrier.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../vgpa_amd/csrc -I../../include pe_product.hip -o pe_product
#include "ode_mfma_impl.h"
#include <cstdio>
using namespace vgpa;
using namespace vgpa::mfma;
template <int MODE>
__global__ void __launch_bounds__(256) k(long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = Geo<10>;
  Lds<10> L; L.carve(smem);
  for (int i = threadIdx.x; i < g::PROB; i += 256) smem[i] = 1e-3 * (i % 97);
  PTab<10> T; build_ptab<10>(40, threadIdx.x >> 6, threadIdx.x & 63, T); double w0[Geo<10>::MAXU] = {};
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    product<10, g::LDA>(L.R, L.X, L.W, T, threadIdx.x & 63, w0, [](){});
    if (MODE == 1) __syncthreads();
    if (MODE == 2) { __syncthreads(); __syncthreads(); }
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  long long* d; hipMalloc(&d, 8 * 4 * 256);
  const size_t lds = Geo<10>::PROB * 8;
  const int iters = 2000;
  for (int mode = 0; mode < 3; mode++) for (int blocks : {1, 256}) {
    auto kern = mode == 0 ? k<0> : (mode == 1 ? k<1> : k<2>);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, iters); hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, iters); hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("mode %d (%s) blocks %3d: cycles per product: %.0f %.0f %.0f %.0f  (MAXU=%d: %d MFMAs, %d cycles of issue)  %s\n", mode,
           mode == 0 ? "no barrier" : (mode == 1 ? "one barrier" : "two barriers"), blocks, (double)h[0] / iters, (double)h[1] / iters,
           (double)h[2] / iters, (double)h[3] / iters, Geo<10>::MAXU, Geo<10>::MAXU * 10, Geo<10>::MAXU * 160, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
