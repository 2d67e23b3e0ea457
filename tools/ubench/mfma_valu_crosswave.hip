// Two waves of ONE SIMD (waves w and w + 4 of a 512-thread workgroup): wave A streams v_mfma_f64_4x4x4_4b products on independent
// accumulators, wave B streams plain VALU work (32-bit integer adds, fp64 adds, LDS reads).  How much does each slow the other, and
// does s_setprio on B change it?  Cycles (s_memtime) per instruction of each role, alone and together.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_crosswave.hip -o mfma_valu_crosswave
#include <hip/hip_runtime.h>
#include <cstdio>
// KIND: 0 = v_add_u32, 1 = v_add_f64, 2 = ds_read_b64 (+ dependent add), 3 = v_cndmask-like select chain
template <int KIND>
__global__ void __launch_bounds__(512) k(double* out, long long* cyc, int iters, int run_a, int run_b, int prio_b) {
  __shared__ double lds[1024];
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.0;
  __syncthreads();
  double res = 0.0;
  long long t0 = 0, t1 = 0;
  if (wave < 4) {
    if (run_a) {
      double a0 = threadIdx.x * 1e-3, b0 = a0 + 2.0;
      double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      t0 = clock64();
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) w[m & 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, w[m & 7], 0, 0, 0);
      }
      t1 = clock64();
      for (int m = 0; m < 8; m++) res += w[m];
    }
  } else if (run_b) {
    if (prio_b) __builtin_amdgcn_s_setprio(3);
    int r[8]; double d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { r[i] = threadIdx.x + i; d[i] = 1.0 + i; }
    t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int m = 0; m < 16; m++) {
        if (KIND == 0) r[m & 7] = r[m & 7] + r[(m + 1) & 7];
        if (KIND == 1) d[m & 7] = d[m & 7] + d[(m + 1) & 7];
        if (KIND == 2) d[m & 7] = lds[(r[m & 7] + m) & 1023];
        if (KIND == 3) r[m & 7] = (r[(m + 1) & 7] > m) ? r[m & 7] : r[(m + 3) & 7] + 1;
      }
      if (KIND == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
    t1 = clock64();
    for (int i = 0; i < 8; i++) res += r[i] + d[i];
  }
  out[threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}
template <int KIND>
void run(const char* name, double* out, long long* cyc) {
  const int iters = 20000;
  for (int mode = 0; mode < 4; mode++) {
    const int ra = mode != 1, rb = mode != 0, pb = mode == 3;
    long long h[2] = {0, 0};
    (void)hipMemset(cyc, 0, 16);
    hipLaunchKernelGGL((k<KIND>), dim3(1), dim3(512), 0, 0, out, cyc, iters, ra, rb, pb);
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-12s %-22s: MFMA wave %.2f cycles per product | VALU wave %.2f cycles per instruction\n", name,
           mode == 0 ? "MFMA alone" : mode == 1 ? "VALU alone" : mode == 2 ? "together" : "together, VALU prio 3", (double)h[0] / (16.0 * iters), (double)h[1] / (16.0 * iters));
  }
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&cyc, 16);
  run<0>("v_add_u32", out, cyc); run<1>("v_add_f64", out, cyc); run<2>("ds_read_b64", out, cyc); run<3>("cmp+cndmask", out, cyc);
  return 0;
}
