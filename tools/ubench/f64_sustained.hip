// Micro-benchmark: what the fp64 matrix pipe SUSTAINS on a whole MI355X under its power cap -- the practical counterpart of the
// nominal 78.6 TFLOP/s (1024 SIMDs x 32 flop/clk x 2.4 GHz) that bench.py's roofline fractions are priced against.
// Every SIMD of every CU runs back-to-back v_mfma_f64_4x4x4_4b (or 16x16x4) on register operands -- no LDS, no HBM, no vector ALU --
// for `seconds`; prints TFLOP/s from HIP events.  Run it beside `rocm-smi --showclocks --showpower` (tools/sustained_peak.sh).
//   operands: "rand" = one random pair per lane, "rand8" = eight pairs used in turn (what a real kernel does), "zero" = zeros
// Build: hipcc --offload-arch=gfx950 -O3 f64_sustained.hip -o f64_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITER = 1 << 16;          // products per accumulator and launch

template <int SHAPE, int NOP>          // SHAPE 0: 4x4x4_4b (512 flop), 1: 16x16x4 (2048 flop)
__global__ void __launch_bounds__(256) k_sustain(double* out, const double* in, int iters) {
  // NOP operand pairs per lane, used in turn: with ONE pair (NOP = 1) consecutive products see the same A / B registers and only the
  // accumulators switch; a real kernel feeds different fragments to every product (NOP = 8)
  double x[NOP], y[NOP];
#pragma unroll
  for (int q = 0; q < NOP; q++) { x[q] = in[(threadIdx.x + 37 * q) & 255]; y[q] = in[256 + ((threadIdx.x + 91 * q) & 255)]; }
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; it += (NOP + 3) / 4) {
#pragma unroll
    for (int q0 = 0; q0 < NOP; q0 += 4) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int q = (q0 + i) % NOP;
        if (SHAPE == 0) s[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[q], y[q], s[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[q], y[q], acc[i], 0, 0, 0);
      }
    }
  }
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) r += s[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 0;
  const int wg_per_cu = argc > 2 ? atoi(argv[2]) : 2;               // 256-thread workgroups per CU (waves per SIMD)
  const double seconds = argc > 3 ? atof(argv[3]) : 4.0;
  const bool zero = argc > 4 && !strcmp(argv[4], "zero");
  const bool many = argc > 4 && !strcmp(argv[4], "rand8");        // eight operand pairs per lane in turn
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * wg_per_cu;
  double *out, *in; CK(hipMalloc(&out, sizeof(double) * 256 * blocks)); CK(hipMalloc(&in, sizeof(double) * 512));
  double h[512];
  srand(1);
  for (int i = 0; i < 512; i++) h[i] = zero ? 0.0 : (0.5 + 0.5 * rand() / RAND_MAX) * (i & 1 ? 1e-3 : -1e-3);   // small: the sums stay finite
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (shape == 0 && many) hipLaunchKernelGGL((k_sustain<0, 8>), dim3(blocks), dim3(256), 0, 0, out, in, ITER);
    else if (shape == 0) hipLaunchKernelGGL((k_sustain<0, 1>), dim3(blocks), dim3(256), 0, 0, out, in, ITER);
    else if (many) hipLaunchKernelGGL((k_sustain<1, 8>), dim3(blocks), dim3(256), 0, 0, out, in, ITER);
    else hipLaunchKernelGGL((k_sustain<1, 1>), dim3(blocks), dim3(256), 0, 0, out, in, ITER);
  };
  launch(); CK(hipDeviceSynchronize());
  const double flop_per_launch = (double)blocks * 4 /*waves*/ * 4 /*acc*/ * ITER * (shape == 0 ? 512.0 : 2048.0);
  double total_ms = 0.0; int n = 0;
  while (total_ms < 1e3 * seconds) {
    CK(hipEventRecord(e0));
    for (int k = 0; k < 8; k++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms; n += 8;
    if ((int)(total_ms / 500.0) != (int)((total_ms - ms) / 500.0))
    printf("%s %s, %d workgroup(s) of 4 waves per CU: %.2f TFLOP/s (last 8 launches, %.1f ms)\n", shape == 0 ? "v_mfma_f64_4x4x4_4b" : "v_mfma_f64_16x16x4",
           zero ? "zero operands" : (many ? "8 random operand pairs in turn" : "random operands"), wg_per_cu, 8 * flop_per_launch / (1e-3 * ms) / 1e12, ms);
    fflush(stdout);
  }
  printf("SUSTAINED %s %s wg/cu=%d: %.2f TFLOP/s over %.1f s\n", shape == 0 ? "4x4x4_4b" : "16x16x4", zero ? "zero" : (many ? "rand8" : "rand"), wg_per_cu,
         n * flop_per_launch / (1e-3 * total_ms) / 1e12, 1e-3 * total_ms);
  return 0;
}
