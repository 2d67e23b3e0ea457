// Micro-benchmark: fp64 issue rates on gfx950 (cycles per wave-instruction on one SIMD).
//   (a) v_mfma_f64_16x16x4_f64   (b) v_mfma_f64_4x4x4_4b_f64   (c) v_fma_f64
//   (d) MFMA wave + VALU wave co-resident on one SIMD (do the fp64 pipes overlap?)
// Build: hipcc --offload-arch=gfx950 -O3 f64_rates.hip -o f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITER = 2048;

template <int MODE, int NACC>
__global__ void __launch_bounds__(1024) k_rate(double* out, long long* cyc, int waves_mfma) {
  const int wave = threadIdx.x >> 6;
  double x = 1.0 + 1e-9 * threadIdx.x, y = 1.0 - 1e-9 * threadIdx.x;
  d4 acc[NACC];
  double s[NACC * 4];
#pragma unroll
  for (int i = 0; i < NACC; i++) { acc[i] = d4{0, 0, 0, 0}; }
#pragma unroll
  for (int i = 0; i < NACC * 4; i++) s[i] = 0.0;
  __syncthreads();
  long long t0 = clock64();
  const bool do_mfma = (MODE == 0) || (MODE == 1) || (MODE == 3 && wave < waves_mfma);
  if (do_mfma) {
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        if (MODE == 1) {
          s[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, s[i], 0, 0, 0);
        } else {
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
        }
      }
    }
  } else {
    for (int it = 0; it < ITER; it++) {
#pragma unroll
      for (int i = 0; i < NACC * 4; i++) s[i] = __builtin_fma(x, y, s[i]);
    }
  }
  long long t1 = clock64();
  double r = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < NACC * 4; i++) r += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int MODE, int NACC>
int run(const char* name, int threads, int blocks, int waves_mfma, double per_wave_instr) {
  double* out; long long* cyc;
  CK(hipMalloc(&out, sizeof(double) * threads * blocks));
  CK(hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_rate<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, waves_mfma);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_rate<MODE, NACC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, waves_mfma);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(blocks * (threads / 64));
  CK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  printf("%-44s thr=%4d blk=%4d  ms=%.4f  ", name, threads, blocks, ms);
  for (int w = 0; w < threads / 64 && w < 8; w++)
    printf(" w%d: %.2f", w, (double)h[w] / (ITER * per_wave_instr));
  printf("  [cycles per instruction per wave]\n");
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  // one wave on one SIMD
  run<0, 1>("mfma_f64_16x16x4, 1 acc (dep chain)", 64, 1, 0, 1);
  run<0, 4>("mfma_f64_16x16x4, 4 acc", 64, 1, 0, 4);
  run<0, 4>("mfma_f64_16x16x4, 4 acc, 4 waves(1/SIMD)", 256, 1, 0, 4);
  run<0, 4>("mfma_f64_16x16x4, 4 acc, 8 waves(2/SIMD)", 512, 1, 0, 4);
  run<0, 4>("mfma_f64_16x16x4, 4 acc, all CUs", 256, 256, 0, 4);
  run<1, 1>("mfma_f64_4x4x4_4b, 1 acc (dep chain)", 64, 1, 0, 1);
  run<1, 4>("mfma_f64_4x4x4_4b, 4 acc", 64, 1, 0, 4);
  run<1, 4>("mfma_f64_4x4x4_4b, 4 acc, 4 waves", 256, 1, 0, 4);
  run<2, 1>("v_fma_f64 x4 chains", 64, 1, 0, 4);
  run<2, 4>("v_fma_f64 x16 chains", 64, 1, 0, 16);
  run<2, 4>("v_fma_f64 x16 chains, 4 waves(1/SIMD)", 256, 1, 0, 16);
  run<2, 4>("v_fma_f64 x16 chains, 8 waves(2/SIMD)", 512, 1, 0, 16);
  run<2, 4>("v_fma_f64 x16 chains, all CUs", 256, 256, 0, 16);
  // co-issue: 8 waves, first 4 MFMA (one per SIMD), last 4 VALU (one per SIMD): per-wave cycles
  // are normalised by 4 instr (mfma waves) -- VALU waves show cycles per 4 fma (x4 to compare).
  run<3, 4>("mixed: 4 MFMA waves + 4 VALU(16 chain) waves", 512, 1, 4, 4);
  return 0;
}
