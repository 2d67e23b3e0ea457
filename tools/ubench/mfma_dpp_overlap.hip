// Do 32-bit VALU moves (v_mov_b32_dpp row_ror:4) issue in the shadow of running v_mfma_f64_4x4x4_4b products of the same wave?
// One wave per SIMD, 14 products per iteration on four accumulators (the stepper's product step), with 0 / 8 / 16 DPP moves whose
// results feed the NEXT iteration's products (so they cannot be dropped) placed between the products.  Cycles per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NDPP, int NWAVE>
__global__ void __launch_bounds__(64 * NWAVE) k(double* out, long long* cyc, int iters) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1.0, b0 = a0 + 2.0, b1 = a0 + 3.0;
  double w[4] = {0.0, 0.0, 0.0, 0.0};
  int r[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 14; m++) {
      const double x = (m & 1) ? a1 : a0, y = (m & 2) ? b1 : b0;
      w[m & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, w[m & 3], 0, 0, 0);
      if (m < NDPP / 2 + (NDPP & 1)) {
        r[2 * m] = __builtin_amdgcn_mov_dpp(r[2 * m], 0x124, 0xf, 0xf, false);
        if (2 * m + 1 < NDPP) r[2 * m + 1] = __builtin_amdgcn_mov_dpp(r[2 * m + 1], 0x124, 0xf, 0xf, false);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  int s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = w[0] + w[1] + w[2] + w[3] + s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NDPP, int NWAVE>
void run(double* out, long long* cyc) {
  const int iters = 20000;
  long long h = 0;
  hipLaunchKernelGGL((k<NDPP, NWAVE>), dim3(1), dim3(64 * NWAVE), 0, 0, out, cyc, iters);
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("waves/SIMD %d  dpp moves/iteration %2d : %.1f clock ticks per iteration of 14 products\n", NWAVE / 4, NDPP, (double)h / iters);
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&cyc, 8);
  run<0, 4>(out, cyc); run<8, 4>(out, cyc); run<16, 4>(out, cyc);
  run<0, 8>(out, cyc); run<8, 8>(out, cyc); run<16, 8>(out, cyc);
  printf("(s_memtime / readcyclecounter ticks at 100 MHz on this part: compare rows, not absolute numbers)\n");
  return 0;
}
