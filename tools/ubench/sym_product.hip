// Micro-benchmark for the symmetric-unit stage (round 2): Z = Aop^T X + X^T Aop on 2x2-block units of the upper triangle, the
// element-wise update and the direct + mirror publish behind the products, ONE barrier per stage -- nothing else (no operand
// staging, no vector recursion, no HBM).  Variants: RL = units per run that share their a-fragments (fragment reads per
// MFMA 1.0 / 0.75 / 0.625), PD = prefetch distance of the fragments in k-pairs; 1, 2 and 3 workgroups per CU.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off sym_product.hip -o sym_product
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr int NB = 10, P = 40, KKE = 10, RP = 2 * KKE, NKP = KKE / 2, LDX = 80, NSB = 5, NU = 15, MAXS = 4;
constexpr int XS = RP * LDX;

template <int RL, int MODE, int PD>   // PD: prefetch distance in k-pairs (1: two fragment buffers, 2: three); RL: slots [r*RL, (r+1)*RL) share their a fragments; MODE 1: update + publish + barrier; 2: barrier only
__global__ void __launch_bounds__(256) k(long long* out, double* sink, int iters, int pad_lds) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* X0 = smem; double* X1 = X0 + XS; double* R = X1 + XS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * XS; i += 256) smem[i] = 1e-3 * (i % 97);
  const int r4 = lane >> 4, b = (lane >> 2) & 3, c4 = lane & 3, bi = b >> 1, bj = b & 1;
  constexpr int NA = MAXS / RL;
  int colI[NA], colJ[MAXS], offD[MAXS], offM[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; s++) {
    int u = wave + 4 * s; if (u >= NU) u = NU - 1;
    int I0 = 0, rem = u;
    while (rem >= NSB - I0) { rem -= NSB - I0; I0++; }
    const int J0 = I0 + rem;
    const int Ib = 2 * I0 + bi, Jb = 2 * J0 + bj;
    if (s % RL == 0) colI[s / RL] = 2 * ((4 * Ib + c4) ^ r4);
    colJ[s] = 2 * ((4 * Jb + c4) ^ r4);
    const int row = 4 * Ib + r4, col = 4 * Jb + c4;
    offD[s] = (row >> 1) * LDX + 2 * (col ^ ((row >> 1) & 3)) + (row & 1);
    offM[s] = (col >> 1) * LDX + 2 * (row ^ ((col >> 1) & 3)) + (col & 1);
  }
  double xk[MAXS], acc[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; s++) { xk[s] = 0.001 * lane; acc[s] = 0.0; }
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    const double* Xc = (it & 1) ? X1 : X0;
    double* Xn = (it & 1) ? X0 : X1;
    const double* pa = R + r4 * LDX;
    const double* px = Xc + r4 * LDX;
    double w[MAXS];
    d2_t a1[PD + 1][NA], a2[PD + 1][NA], b1[PD + 1][MAXS], b2[PD + 1][MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; s++) w[s] = -xk[s];
    auto load = [&](int buf, int kp) {
#pragma unroll
      for (int r = 0; r < NA; r++) {
        a1[buf][r] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LDX + colI[r]);
        a2[buf][r] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LDX + colI[r]);
      }
#pragma unroll
      for (int s = 0; s < MAXS; s++) {
        b1[buf][s] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LDX + colJ[s]);
        b2[buf][s] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LDX + colJ[s]);
      }
    };
    load(0, 0);
    if (PD == 2) load(1, 1);
#pragma unroll
    for (int kp = 0; kp < NKP; kp++) {
      const int cur = kp % (PD + 1);
      if (kp + PD < NKP) load((kp + PD) % (PD + 1), kp + PD);
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int s = 0; s < MAXS; s++) w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[cur][s / RL][h], b1[cur][s][h], w[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < MAXS; s++) w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[cur][s / RL][h], b2[cur][s][h], w[s], 0, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < MAXS; s++) {
      if (MODE == 1) {
        const double f = -w[s];
        acc[s] = acc[s] + 2.0 * f;
        const double xn = xk[s] + 0.005 * f;
        Xn[offD[s]] = xn;
        Xn[offM[s]] = xn;
      } else {
        acc[s] += w[s];
      }
    }
    __syncthreads();
  }
  long long t1 = clock64();
  double sum = 0;
#pragma unroll
  for (int s = 0; s < MAXS; s++) sum += acc[s];
  if (sum == 1.2345) sink[tid] = sum;
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

// Eight-wave variant: one problem per workgroup of EIGHT waves, each wave ONE run of two units (40 MFMAs per stage), so that two
// workgroups per CU put FOUR in-order waves on every SIMD (needs <= 128 registers).
__global__ void __launch_bounds__(512) k8(long long* out, double* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* X0 = smem; double* X1 = X0 + XS; double* R = X1 + XS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * XS; i += 512) smem[i] = 1e-3 * (i % 97);
  const int r4 = lane >> 4, b = (lane >> 2) & 3, c4 = lane & 3, bi = b >> 1, bj = b & 1;
  int colI, colJ[2], offD[2], offM[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    int u = 2 * wave + s; if (u >= NU) u = NU - 1;
    int I0 = 0, rem = u;
    while (rem >= NSB - I0) { rem -= NSB - I0; I0++; }
    const int J0 = I0 + rem;
    const int Ib = 2 * I0 + bi, Jb = 2 * J0 + bj;
    if (s == 0) colI = 2 * ((4 * Ib + c4) ^ r4);
    colJ[s] = 2 * ((4 * Jb + c4) ^ r4);
    const int row = 4 * Ib + r4, col = 4 * Jb + c4;
    offD[s] = (row >> 1) * LDX + 2 * (col ^ ((row >> 1) & 3)) + (row & 1);
    offM[s] = (col >> 1) * LDX + 2 * (row ^ ((col >> 1) & 3)) + (col & 1);
  }
  double xk[2], acc[2];
#pragma unroll
  for (int s = 0; s < 2; s++) { xk[s] = 0.001 * lane; acc[s] = 0.0; }
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    const double* Xc = (it & 1) ? X1 : X0;
    double* Xn = (it & 1) ? X0 : X1;
    const double* pa = R + r4 * LDX;
    const double* px = Xc + r4 * LDX;
    double w[2];
    d2_t a1[2], a2[2], b1[2][2], b2[2][2];
#pragma unroll
    for (int s = 0; s < 2; s++) w[s] = -xk[s];
    auto load = [&](int buf, int kp) {
      a1[buf] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LDX + colI);
      a2[buf] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LDX + colI);
#pragma unroll
      for (int s = 0; s < 2; s++) {
        b1[buf][s] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LDX + colJ[s]);
        b2[buf][s] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LDX + colJ[s]);
      }
    };
    load(0, 0);
#pragma unroll
    for (int kp = 0; kp < NKP; kp++) {
      const int cur = kp & 1;
      if (kp + 1 < NKP) load(cur ^ 1, kp + 1);
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int s = 0; s < 2; s++) w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[cur][h], b1[cur][s][h], w[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; s++) w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[cur][h], b2[cur][s][h], w[s], 0, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const double f = -w[s];
      acc[s] = acc[s] + 2.0 * f;
      const double xn = xk[s] + 0.005 * f;
      Xn[offD[s]] = xn;
      Xn[offM[s]] = xn;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  long long t1 = clock64();
  double sum = acc[0] + acc[1];
  if (sum == 1.2345) sink[tid] = sum;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

void run8(long long* d, double* sink) {
  const int iters = 2000;
  const size_t lds = 3 * XS * 8;
  hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int blocks : {256, 512, 768, 1024}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k8, dim3(blocks), dim3(512), lds, 0, d, sink, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k8, dim3(blocks), dim3(512), lds, 0, d, sink, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("eight waves per problem (one run each) blocks=%4d: %.3f ms = %.0f ns per stage-launch = %.0f ns per problem-stage; wave cycles/stage %.0f  %s\n",
           blocks, ms, ms * 1e6 / iters, ms * 1e6 / iters / (blocks / 256.0), (double)h[0] / iters, hipGetErrorString(hipGetLastError()));
  }
}

// Pair variant: ONE workgroup of 8 waves per CU works on TWO problems.  Waves 0-3 multiply -- problem A's stage, then
// problem B's, in one software pipeline (B's first fragments fly during A's last products), each followed by its stepper
// and publish -- waves 4-7 do a synthetic stand-in of everything else (HELP LDS reads + FMAs + writes per problem); one
// barrier per pair of stages.
template <int RL, int HELP>
__global__ void __launch_bounds__(512) k2(long long* out, double* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PS = 4 * XS;                     // per problem: X0, X1, R, M
  for (int i = tid; i < 2 * PS; i += 512) smem[i] = 1e-3 * (i % 97);
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4) {
    const int r4 = lane >> 4, b = (lane >> 2) & 3, c4 = lane & 3, bi = b >> 1, bj = b & 1;
    constexpr int NA = MAXS / RL;
    int colI[NA], colJ[MAXS], offD[MAXS], offM[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; s++) {
      int u = wave + 4 * s; if (u >= NU) u = NU - 1;
      int I0 = 0, rem = u;
      while (rem >= NSB - I0) { rem -= NSB - I0; I0++; }
      const int J0 = I0 + rem;
      const int Ib = 2 * I0 + bi, Jb = 2 * J0 + bj;
      if (s % RL == 0) colI[s / RL] = 2 * ((4 * Ib + c4) ^ r4);
      colJ[s] = 2 * ((4 * Jb + c4) ^ r4);
      const int row = 4 * Ib + r4, col = 4 * Jb + c4;
      offD[s] = (row >> 1) * LDX + 2 * (col ^ ((row >> 1) & 3)) + (row & 1);
      offM[s] = (col >> 1) * LDX + 2 * (row ^ ((col >> 1) & 3)) + (col & 1);
    }
    double xk[2][MAXS], acc[2][MAXS];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int s = 0; s < MAXS; s++) { xk[q][s] = 0.001 * lane; acc[q][s] = 0.0; }
    d2_t a1[2][NA], a2[2][NA], b1[2][MAXS], b2[2][MAXS];
    auto load = [&](int buf, int t, int it) {       // t = q * NKP + kp
      const int q = t / NKP, kp = t % NKP;
      const double* X = smem + q * PS + ((it & 1) ? XS : 0);
      const double* R = smem + q * PS + 2 * XS;
      const double* pa = R + r4 * LDX;
      const double* px = X + r4 * LDX;
#pragma unroll
      for (int r = 0; r < NA; r++) {
        a1[buf][r] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LDX + colI[r]);
        a2[buf][r] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LDX + colI[r]);
      }
#pragma unroll
      for (int s = 0; s < MAXS; s++) {
        b1[buf][s] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LDX + colJ[s]);
        b2[buf][s] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LDX + colJ[s]);
      }
    };
    load(0, 0, 0);
    for (int it = 0; it < iters; it++) {
      double w[MAXS];
#pragma unroll
      for (int t = 0; t < 2 * NKP; t++) {
        const int q = t / NKP, kp = t % NKP, cur = t & 1;
        if (t + 1 < 2 * NKP) load(cur ^ 1, t + 1, it);
        if (kp == 0) {
#pragma unroll
          for (int s = 0; s < MAXS; s++) w[s] = -xk[q][s];
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
          for (int s = 0; s < MAXS; s++) w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[cur][s / RL][h], b1[cur][s][h], w[s], 0, 0, 0);
#pragma unroll
          for (int s = 0; s < MAXS; s++) w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[cur][s / RL][h], b2[cur][s][h], w[s], 0, 0, 0);
        }
        if (kp == NKP - 1) {
          double* Xn = smem + q * PS + ((it & 1) ? 0 : XS);
#pragma unroll
          for (int s = 0; s < MAXS; s++) {
            const double f = -w[s];
            acc[q][s] = acc[q][s] + 2.0 * f;
            const double xn = xk[q][s] + 0.005 * f;
            Xn[offD[s]] = xn;
            Xn[offM[s]] = xn;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      load(0, 0, it + 1);
    }
    double sum = 0;
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int s = 0; s < MAXS; s++) sum += acc[q][s];
    if (sum == 1.2345) sink[tid] = sum;
  } else {
    const int ht = tid - 256;
    double keep = 0.0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const double* M = smem + q * PS + 3 * XS;
        d2_t v[HELP > 0 ? HELP : 1];
#pragma unroll
        for (int r = 0; r < HELP; r++) v[r] = *reinterpret_cast<const d2_t*>(M + 2 * ((ht + 256 * r) % (XS / 2)));
        double s0 = 0.0;
#pragma unroll
        for (int r = 0; r < HELP; r++) s0 = __builtin_fma(v[r][0], v[r][1], s0);
        keep += s0;
#pragma unroll
        for (int r = 0; r < HELP / 2; r++) {
          d2_t o; o[0] = s0; o[1] = 0.5 * s0 + 1e-3;
          *reinterpret_cast<d2_t*>(smem + q * PS + 3 * XS + 2 * ((ht + 256 * r) % (XS / 2))) = o * 1e-3;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (keep == 1.2345) sink[tid] = keep;
  }
  long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RL, int HELP>
void run2(long long* d, double* sink) {
  const int iters = 2000;
  const size_t lds = 2 * 4 * XS * 8;
  auto kern = k2<RL, HELP>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int blocks : {1, 256}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d, sink, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, d, sink, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("pair workgroup (2 problems, 4 product + 4 helper waves) RL=%d HELP=%2d blocks=%3d: %.3f ms = %.0f ns per PAIR of stages; wave cycles: product %.0f helper %.0f  %s\n",
           RL, HELP, blocks, ms, ms * 1e6 / iters, (double)h[0] / iters, (double)h[4] / iters, hipGetErrorString(hipGetLastError()));
  }
}

template <int IL, int MODE, int PD>
void run(long long* d, double* sink, const char* name) {
  const int iters = 2000;
  for (int two : {0, 1}) {
    const size_t lds = two ? 3 * XS * 8 : 100 * 1024;    // 46 KB -> several workgroups per CU; 100 KB -> one
    auto kern = k<IL, MODE, PD>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int blocks : {256, 512, 768}) {
      if (!two && blocks != 256) continue;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, sink, iters, 0); hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, sink, iters, 0);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      printf("%-28s PD=%d RL=%d lds=%3zuKB blocks=%3d: %.3f ms = %.0f ns per stage-launch; wave cycles/stage %.0f %.0f %.0f %.0f  %s\n", name, PD, IL, lds / 1024, blocks, ms,
             ms * 1e6 / iters, (double)h[0] / iters, (double)h[1] / iters, (double)h[2] / iters, (double)h[3] / iters, hipGetErrorString(hipGetLastError()));
    }
  }
}

int main() {
  long long* d; hipMalloc(&d, 8 * 8 * 1024);
  double* sink; hipMalloc(&sink, 8 * 512);
  run<2, 1, 1>(d, sink, "product+update+publish+bar");
  run8(d, sink);
  run2<2, 0>(d, sink);
  run2<2, 8>(d, sink);
  run2<2, 16>(d, sink);
  run2<4, 8>(d, sink);
  return 0;
}
