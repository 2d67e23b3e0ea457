// Symmetric fast path of the time-stepping kernels: fp64 matrix cores, one workgroup per problem.
// (Included by ode_mfma_m{0,1,2,3}.hip, one translation unit per stepper so that they compile in parallel.)
//
// Math.  With S (resp. Psi) symmetric the two products of the reference collapse to one:
//     forward : f_S   = -A S - S A^T + Sigma = -(W + W^T) + Sigma,   W  = A S        (ode_solver.py:60)
//     backward: f_Psi = -G + Psi A + A^T Psi = -G + W'^T + W',       W' = A^T Psi    (ode_solver.py:94)
// so a stage costs ONE D^3 product.  The steppers are those of src/numerics/{euler,heun,runge_kutta2,
// runge_kutta4}.py (incl. the RK2 covariance predictor that passes S_k as A, runge_kutta2.py:96, and
// f_lam = -g + A.lam, ode_solver.py:77); jumps are added after the step (euler.py:139-149).
//
// Mapping to gfx950.  The product runs on v_mfma_f64_4x4x4_4b_f64 (16 cycles, 4 independent 4x4x4 blocks, same
// 16 FMA/clk/SIMD as the 16x16x4 shape -- measured, profiles/r01_fp64_issue_rates.txt) so that D = 40 needs NO
// padding.  Lane l of the instruction holds  A-operand  Aop[4kk + (l>>4)][4 I_b + (l&3)]
//                                            B-operand  X  [4kk + (l>>4)][4 J_b + (l&3)]
//                                            result     W  [4 I_b + (l>>4)][4 J_b + (l&3)],   b = (l>>2)&3
// (layout probed on hardware, profiles/r01_fp64_mfma_layout_probe.txt).  A "unit" is one MFMA accumulator = four
// 4x4 output blocks: (I, J = 4q..4q+3) for the full column groups, and the left-over column blocks of several
// block-rows packed together, so the 100 blocks of a 40x40 product make exactly 25 units = 250 MFMAs per stage,
// dealt 6/6/6/7 to the four waves (one per SIMD).  Operands are read from LDS with immediate offsets (the k loop is
// fully unrolled): the stage state X row-major with a leading dimension = 16 (mod 32) doubles (conflict-free
// 16-wide rows), the A operand (A^T forward, A backward) with an odd leading dimension (conflict-free for the
// 4-wide reads and for the column reads of the mat-vec).  Fragments are double-buffered in registers.
// Each lane OWNS the W elements its accumulators hold: S_k / Psi_t, the Runge-Kutta sums, Sigma and G live in its
// registers; W^T is obtained through one LDS exchange per stage.  A_{k+2} is prefetched from HBM one step ahead.
#pragma once
#include "vgpa_internal.h"

namespace vgpa {
namespace mfma {

constexpr int NT = 256;
constexpr int NW = 4;
constexpr int kMaxNB = 11;   // D <= 44: beyond that the backward kernel spills registers (generic path instead)

// Compile-time geometry of the padded problem: NB = ceil(D/4) 4x4 blocks per dimension.
template <int NB_>
struct Geo {
  static constexpr int NB = NB_;
  static constexpr int NQ = NB / 4;                       // full 16-column groups
  static constexpr int REM = NB % 4;                      // left-over column blocks per block-row
  static constexpr int G = REM ? 4 / REM : 0;             // block-rows packed into one left-over unit
  static constexpr int NLEFT = REM ? (NB + G - 1) / G : 0;
  static constexpr int NU = NB * NQ + NLEFT;              // units (MFMA accumulators) per product
  static constexpr int MAXU = (NU + NW - 1) / NW;         // unit slots per wave
  static constexpr int P = 4 * NB;                        // padded dimension
  static constexpr int PC = 16 * ((P + 15) / 16);
  static constexpr int LDX = (PC % 32 == 16) ? PC : PC + 16;   // = 16 (mod 32): conflict-free 16-wide rows
  // Leading dimension of the A operand buffers.  Odd and = 5 (mod 32): conflict-free for the 4-wide fragment
  // reads, the column reads of the mat-vec and the transposing stores; 4*LDA*8 B = 2208 B is > 2040 B and not a
  // multiple of 512 B, which keeps hipcc from fusing the reads of two k-steps into ds_read2(st64)_b64 (half the
  // LDS rate of ds_read_b64, MI355X_MICROARCH.md s.LDS).
  static constexpr int LDA = 69;
  static constexpr int LDW = P + 1;                       // exchange buffer for W^T (odd)
  static constexpr int KKE = NB + (NB & 1);               // k-steps rounded up to even (extra rows are zero)
  static constexpr int ROWS = 4 * KKE;                    // rows allocated per LDS matrix (zero padded)
  static constexpr int EPT = (P * P + NT - 1) / NT;       // A entries per thread for the HBM -> LDS staging
  static constexpr size_t LDS_DOUBLES = (size_t)ROWS * LDX + (size_t)P * LDW + 3 * (size_t)ROWS * LDA +
                                        (size_t)(2 + NW) * P + 8;
};

template <int NB>
struct Lds {
  double* X;     // [ROWS][LDX]  stage state
  double* W;     // [P][LDW]     exchange buffer for W^T
  double* A0;    // [ROWS][LDA]  operand of A at the step's start point
  double* AM;    // [ROWS][LDA]  operand of the mid-point
  double* A1;    // [ROWS][LDA]  operand of A at the step's end point
  double* xv;    // [P]          stage vector (m or lam)
  double* pv;    // [NW][P]      partial mat-vec sums
  __device__ __forceinline__ void carve(double* smem) {
    using g = Geo<NB>;
    X = smem; W = X + g::ROWS * g::LDX; A0 = W + g::P * g::LDW; AM = A0 + g::ROWS * g::LDA;
    A1 = AM + g::ROWS * g::LDA; xv = A1 + g::ROWS * g::LDA; pv = xv + 2 * g::P;
  }
};

template <int NB>
struct Tab {
  static constexpr int MAXU = Geo<NB>::MAXU;
  int colA[MAXU];   // 4*I_b + (l&3)
  int colB0, colB1; // B-fragment columns of the wave's first / second column group
  int offWw[MAXU];  // row*LDW + col
  int offWr[MAXU];  // col*LDW + row
  int offX[MAXU];   // row*LDX + col
  int gofs[MAXU];   // row*D + col   (global element offset inside a D x D matrix)
  unsigned valid;   // per-lane bit s: this lane owns a real matrix element in slot s
  int nfirst;       // wave-uniform: slots [0, nfirst) use the first B fragment, the others the second
};

template <int NB>
__device__ __forceinline__ void build_tab(int D, Tab<NB>& T) {
  using g = Geo<NB>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = (lane >> 2) & 3, r4 = lane >> 4, c4 = lane & 3;
  const int u0 = (wave * g::NU) / NW, u1 = ((wave + 1) * g::NU) / NW;
  T.valid = 0u;
  T.colB0 = T.colB1 = c4;
  int first_group = -1, nfirst = 0;
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) {
    const int u = u0 + s;
    int Ib = 0, Jb = 0, group = first_group;
    bool ok = false;
    if (u < u1) {
      if (u < g::NB * g::NQ) {
        const int q = u / g::NB;
        Ib = u - q * g::NB; Jb = 4 * q + b; group = q; ok = true;
      } else {
        constexpr int rem = g::REM ? g::REM : 1;
        const int v = u - g::NB * g::NQ;
        const int i0 = v * g::G;
        Ib = i0 + b / rem; Jb = 4 * g::NQ + b % rem; group = g::NQ;
        ok = (b < g::G * g::REM) && (Ib < g::NB);
        if (!ok) { Ib = i0; Jb = 4 * g::NQ; }
      }
      if (s == 0) { first_group = group; T.colB0 = T.colB1 = 4 * Jb + c4; }
      if (group == first_group) nfirst = s + 1;
      else T.colB1 = 4 * Jb + c4;
    }
    const int row = 4 * Ib + r4, col = 4 * Jb + c4;
    T.colA[s] = 4 * Ib + c4;
    T.offWw[s] = row * g::LDW + col;
    T.offWr[s] = col * g::LDW + row;
    T.offX[s] = row * g::LDX + col;
    T.gofs[s] = row * D + col;
    if (ok && row < D && col < D) T.valid |= (1u << s);
  }
  T.nfirst = __builtin_amdgcn_readfirstlane(nfirst);
}

// ---- one D^3 product on the matrix cores: w[s] = sum_kk Aop-block x X-block ---------------------------------
// LDAOP = leading dimension of the A-operand matrix (LDA, or LDX when the stage state itself is the operand).
// Straight-line code: KKE k-steps, fragments of step kk+1 are loaded while the MFMAs of step kk issue.
template <int NB, int LDAOP>
__device__ __forceinline__ void mfma_product(const double* __restrict__ Aop, const double* __restrict__ X,
                                             const Tab<NB>& T, double (&w)[Geo<NB>::MAXU]) {
  using g = Geo<NB>;
  constexpr int MAXU = g::MAXU;
  const int r4 = (threadIdx.x & 63) >> 4;
  const double* pa = Aop + r4 * LDAOP;
  const double* px = X + r4 * g::LDX;
  double af[2][MAXU], bf[2][2];
#pragma unroll
  for (int s = 0; s < MAXU; s++) { w[s] = 0.0; af[0][s] = pa[T.colA[s]]; }
  bf[0][0] = px[T.colB0];
  bf[0][1] = px[T.colB1];
#pragma unroll
  for (int kk = 0; kk < g::KKE; kk++) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < g::KKE) {
#pragma unroll
      for (int s = 0; s < MAXU; s++) af[nxt][s] = pa[(kk + 1) * 4 * LDAOP + T.colA[s]];
      bf[nxt][0] = px[(kk + 1) * 4 * g::LDX + T.colB0];
      bf[nxt][1] = px[(kk + 1) * 4 * g::LDX + T.colB1];
    }
#pragma unroll
    for (int s = 0; s < MAXU; s++) {
      const double b = (s < T.nfirst) ? bf[cur][0] : bf[cur][1];
      w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[cur][s], b, w[s], 0, 0, 0);
    }
  }
}

// partial mat-vec of this wave (k-range = its quarter): forward sum_k Aop[k][i] v[k], backward sum_k Aop[i][k] v[k]
template <int NB, bool FWD>
__device__ __forceinline__ double matvec_partial(const double* __restrict__ Aop, const double* __restrict__ xv, int D) {
  using g = Geo<NB>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int KQ = (g::P + NW - 1) / NW;     // rows of padding are zero, so the padded range is harmless
  const int k0 = wave * KQ;
  const int li = (lane < g::P) ? lane : 0;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < KQ; k++) {
    if (k0 + k < g::P) {
      const double av = FWD ? Aop[(k0 + k) * g::LDA + li] : Aop[li * g::LDA + (k0 + k)];
      s = __builtin_fma(av, xv[k0 + k], s);
    }
  }
  (void)D;
  return s;
}

// Products of one stage + the LDS exchange.  On return: w = own W element, wt = W^T element, vsum = (Aop-matvec)
// for lanes < D of wave 0.  Contains ONE barrier.
template <int NB, bool FWD, int LDAOP>
__device__ __forceinline__ void stage_products(const Lds<NB>& L, int D, const double* Aop, const Tab<NB>& T,
                                               const double* Avec, double (&w)[Geo<NB>::MAXU],
                                               double (&wt)[Geo<NB>::MAXU], double& vsum) {
  using g = Geo<NB>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  mfma_product<NB, LDAOP>(Aop, L.X, T, w);
  const double part = matvec_partial<NB, FWD>(Avec, L.xv, D);
#pragma unroll
  for (int s = 0; s < g::MAXU; s++)
    if ((T.valid >> s) & 1u) L.W[T.offWw[s]] = w[s];
  if (lane < g::P) L.pv[wave * g::P + lane] = part;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) wt[s] = ((T.valid >> s) & 1u) ? L.W[T.offWr[s]] : 0.0;
  vsum = 0.0;
  if (wave == 0 && lane < D)
    vsum = ((L.pv[lane] + L.pv[g::P + lane]) + L.pv[2 * g::P + lane]) + L.pv[3 * g::P + lane];
}

// publish the next stage state (matrix elements owned by this lane + vector entries of wave 0).  ONE barrier.
template <int NB>
__device__ __forceinline__ void publish(const Lds<NB>& L, int D, const Tab<NB>& T,
                                        const double (&xn)[Geo<NB>::MAXU], double vn) {
#pragma unroll
  for (int s = 0; s < Geo<NB>::MAXU; s++)
    if ((T.valid >> s) & 1u) L.X[T.offX[s]] = xn[s];
  if ((threadIdx.x >> 6) == 0 && (threadIdx.x & 63) < D) L.xv[threadIdx.x & 63] = vn;
  __syncthreads();
}

// A(t) from HBM into registers, coalesced (thread e <-> A[e / D][e % D])
template <int NB>
__device__ __forceinline__ void load_a(const double* __restrict__ A, int DD, double (&a)[Geo<NB>::EPT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB>::EPT; q++) {
    const int e = threadIdx.x + q * NT;
    a[q] = (e < DD) ? A[e] : 0.0;
  }
}

// registers -> LDS operand buffer: forward stores A^T (Aop[c][r] = A[r][c]), backward stores A.
template <int NB, bool FWD, bool MID>
__device__ __forceinline__ void store_a(double* __restrict__ buf, int D, const int (&aofs)[Geo<NB>::EPT],
                                        const double (&a0)[Geo<NB>::EPT], const double (&a1)[Geo<NB>::EPT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB>::EPT; q++) {
    if (aofs[q] >= 0) buf[aofs[q]] = MID ? 0.5 * (a0[q] + a1[q]) : a0[q];
  }
  (void)D;
}

template <int NB, bool FWD>
__device__ __forceinline__ void build_aofs(int D, int (&aofs)[Geo<NB>::EPT]) {
  using g = Geo<NB>;
#pragma unroll
  for (int q = 0; q < g::EPT; q++) {
    const int e = threadIdx.x + q * NT;
    const int r = e / D, c = e - r * D;
    aofs[q] = (e < D * D) ? (FWD ? (c * g::LDA + r) : (r * g::LDA + c)) : -1;
  }
}

// =================================================================================================================
template <int METHOD, int NB>
__global__ void __launch_bounds__(NT) k_fwd_mfma(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = Geo<NB>;
  constexpr int MAXU = g::MAXU, EPT = g::EPT;
  const int D = a.D, DD = D * D, Np = a.Np;
  const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Lds<NB> L;
  L.carve(smem);
  const double* A = a.A + (size_t)prob * Np * DD;
  const double* bb = a.b + (size_t)prob * Np * D;
  double* mt = a.m + (size_t)prob * Np * D;
  double* st = a.S + (size_t)prob * Np * DD;
  const double dt = a.dt, h = 0.5 * a.dt;
  const bool vlane = (wave == 0) && (lane < D);

  Tab<NB> T;
  build_tab<NB>(D, T);
  int aofs[EPT];
  build_aofs<NB, true>(D, aofs);
  for (int i = tid; i < (int)g::LDS_DOUBLES; i += NT) smem[i] = 0.0;
  __syncthreads();

  double sk[MAXU], sig[MAXU], w[MAXU], wt[MAXU], r[MAXU], acc1[MAXU], acc2[MAXU], xn[MAXU];
  double aC[EPT], aN[EPT];
  double mk = 0.0, vs = 0.0;
#pragma unroll
  for (int s = 0; s < MAXU; s++) {
    const bool ok = (T.valid >> s) & 1u;
    sk[s] = ok ? a.S0[T.gofs[s]] : 0.0;
    sig[s] = ok ? a.Sigma[T.gofs[s]] : 0.0;
    acc1[s] = acc2[s] = 0.0;
    if (ok) { st[T.gofs[s]] = sk[s]; L.X[T.offX[s]] = sk[s]; }
  }
  if (vlane) { mk = a.m0[lane]; mt[lane] = mk; L.xv[lane] = mk; }
  load_a<NB>(A, DD, aC);
  store_a<NB, true, false>(L.A0, D, aofs, aC, aC);
  if (Np > 1) load_a<NB>(A + DD, DD, aN);
  __syncthreads();

  for (int k = 0; k < Np - 1; k++) {
    // operands of this step: A1 <- A_{k+1}, AM <- mid-point; prefetch A_{k+2} for the next step
    store_a<NB, true, false>(L.A1, D, aofs, aN, aN);
    if (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4) store_a<NB, true, true>(L.AM, D, aofs, aC, aN);
#pragma unroll
    for (int q = 0; q < EPT; q++) aC[q] = aN[q];
    if (k + 2 < Np) load_a<NB>(A + (size_t)(k + 2) * DD, DD, aN);
    const double b0 = vlane ? bb[(size_t)k * D + lane] : 0.0;
    const double b1 = vlane ? bb[(size_t)(k + 1) * D + lane] : 0.0;
    double mnew = 0.0;

    if (METHOD == VGPA_ODE_EULER) {
      stage_products<NB, true, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs);
#pragma unroll
      for (int s = 0; s < MAXU; s++) sk[s] = sk[s] + ((-w[s] - wt[s]) + sig[s]) * dt;
      mnew = mk + (-vs + b0) * dt;
    } else if (METHOD == VGPA_ODE_HEUN) {
      stage_products<NB, true, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs);
      const double pm = -vs + b0;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-w[s] - wt[s]) + sig[s]; xn[s] = sk[s] + acc1[s] * dt; }
      publish<NB>(L, D, T, xn, mk + pm * dt);
      stage_products<NB, true, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs);
      const double cm = -vs + b1;
#pragma unroll
      for (int s = 0; s < MAXU; s++) sk[s] = sk[s] + h * (acc1[s] + ((-w[s] - wt[s]) + sig[s]));
      mnew = mk + h * (pm + cm);
    } else if (METHOD == VGPA_ODE_RK2) {
      // covariance predictor: S_k stands in for A_k (Q2): operand = X itself (S symmetric); mean predictor: A_k
      stage_products<NB, true, g::LDX>(L, D, L.X, T, L.A0, w, wt, vs);
      const double pm = -vs + b0;
#pragma unroll
      for (int s = 0; s < MAXU; s++) xn[s] = sk[s] + h * ((-w[s] - wt[s]) + sig[s]);
      publish<NB>(L, D, T, xn, mk + h * pm);
      stage_products<NB, true, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs);
      const double cm = -vs + 0.5 * (b0 + b1);
#pragma unroll
      for (int s = 0; s < MAXU; s++) sk[s] = sk[s] + dt * ((-w[s] - wt[s]) + sig[s]);
      mnew = mk + dt * cm;
    } else {  // RK4
      const double bmid = 0.5 * (b0 + b1);
      stage_products<NB, true, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs);
      const double k1 = -vs + b0;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-w[s] - wt[s]) + sig[s]; xn[s] = sk[s] + h * acc1[s]; }
      publish<NB>(L, D, T, xn, mk + h * k1);
      stage_products<NB, true, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs);
      const double k2 = -vs + bmid;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc2[s] = (-w[s] - wt[s]) + sig[s]; xn[s] = sk[s] + h * acc2[s]; }
      publish<NB>(L, D, T, xn, mk + h * k2);
      stage_products<NB, true, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs);
      const double k3 = -vs + bmid;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { r[s] = (-w[s] - wt[s]) + sig[s]; acc2[s] = acc2[s] + r[s]; xn[s] = sk[s] + dt * r[s]; }
      publish<NB>(L, D, T, xn, mk + dt * k3);
      stage_products<NB, true, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs);
      const double k4 = -vs + b1;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        r[s] = (-w[s] - wt[s]) + sig[s];
        sk[s] = sk[s] + dt * (acc1[s] + 2.0 * acc2[s] + r[s]) / 6.0;
      }
      mnew = mk + dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0;
    }
    mk = mnew;
    double* so = st + (size_t)(k + 1) * DD;
#pragma unroll
    for (int s = 0; s < MAXU; s++)
      if ((T.valid >> s) & 1u) so[T.gofs[s]] = sk[s];
    if (vlane) mt[(size_t)(k + 1) * D + lane] = mk;
    publish<NB>(L, D, T, sk, mk);
    // rotate operand buffers: A_{k+1} becomes the start-point operand of the next step
    double* tmp = L.A0; L.A0 = L.A1; L.A1 = tmp;
  }
}

// =================================================================================================================
template <int METHOD, int NB>
__global__ void __launch_bounds__(NT) k_bwd_mfma(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = Geo<NB>;
  constexpr int MAXU = g::MAXU, EPT = g::EPT;
  const int D = a.D, DD = D * D, Np = a.Np;
  const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Lds<NB> L;
  L.carve(smem);
  const double* A = a.A + (size_t)prob * Np * DD;
  const double* gm = a.dEm + (size_t)prob * Np * D;
  const double* gs = a.dEs + (size_t)prob * Np * DD;
  double* lam = a.lam + (size_t)prob * Np * D;
  double* psi = a.psi + (size_t)prob * Np * DD;
  const double dt = a.dt, h = 0.5 * a.dt;
  const bool vlane = (wave == 0) && (lane < D);

  Tab<NB> T;
  build_tab<NB>(D, T);
  int aofs[EPT];
  build_aofs<NB, false>(D, aofs);
  for (int i = tid; i < (int)g::LDS_DOUBLES; i += NT) smem[i] = 0.0;
  __syncthreads();

  double pk[MAXU], gC[MAXU], gN[MAXU], jsc[MAXU], w[MAXU], wt[MAXU], r[MAXU], acc1[MAXU], acc2[MAXU], xn[MAXU];
  double aC[EPT], aN[EPT];
  double lk = 0.0, vs = 0.0;
  // here "A0" holds A_t (start point of the backward step), "A1" holds A_{t-1}
#pragma unroll
  for (int s = 0; s < MAXU; s++) {
    const bool ok = (T.valid >> s) & 1u;
    pk[s] = 0.0; acc1[s] = acc2[s] = 0.0;
    gC[s] = ok ? gs[(size_t)(Np - 1) * DD + T.gofs[s]] : 0.0;
    gN[s] = (ok && Np > 1) ? gs[(size_t)(Np - 2) * DD + T.gofs[s]] : 0.0;
    jsc[s] = (ok && a.js_const) ? a.js_const[T.gofs[s]] : 0.0;
    if (ok) psi[(size_t)(Np - 1) * DD + T.gofs[s]] = 0.0;
  }
  if (vlane) lam[(size_t)(Np - 1) * D + lane] = 0.0;
  load_a<NB>(A + (size_t)(Np - 1) * DD, DD, aC);
  store_a<NB, false, false>(L.A0, D, aofs, aC, aC);
  if (Np > 1) load_a<NB>(A + (size_t)(Np - 2) * DD, DD, aN);
  __syncthreads();

  for (int t = Np - 1; t > 0; t--) {
    store_a<NB, false, false>(L.A1, D, aofs, aN, aN);
    if (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4) store_a<NB, false, true>(L.AM, D, aofs, aN, aC);
#pragma unroll
    for (int q = 0; q < EPT; q++) aC[q] = aN[q];
    if (t >= 2) load_a<NB>(A + (size_t)(t - 2) * DD, DD, aN);
    const double g0 = vlane ? gm[(size_t)t * D + lane] : 0.0;          // dEsde_dm[t]
    const double g1 = vlane ? gm[(size_t)(t - 1) * D + lane] : 0.0;    // dEsde_dm[t-1]
    // jumps of index t-1
    double js[MAXU], jm = 0.0;
    if (a.js_dense) {
      const double* jp = a.js_dense + ((size_t)prob * Np + (t - 1)) * DD;
#pragma unroll
      for (int s = 0; s < MAXU; s++) js[s] = ((T.valid >> s) & 1u) ? jp[T.gofs[s]] : 0.0;
      if (vlane) jm = a.jm_dense[((size_t)prob * Np + (t - 1)) * D + lane];
    } else {
      const int n = a.obs_idx ? a.obs_idx[t - 1] : -1;
#pragma unroll
      for (int s = 0; s < MAXU; s++) js[s] = (n >= 0) ? jsc[s] : 0.0;
      if (vlane && n >= 0) jm = a.jm_sparse[((size_t)prob * a.n_obs + n) * D + lane];
    }
    double lnew = 0.0;

    if (METHOD == VGPA_ODE_EULER) {
      stage_products<NB, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs);
#pragma unroll
      for (int s = 0; s < MAXU; s++) pk[s] = pk[s] - ((-gC[s] + wt[s]) + w[s]) * dt + js[s];
      lnew = lk - (-g0 + vs) * dt + jm;
    } else if (METHOD == VGPA_ODE_HEUN) {
      stage_products<NB, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs);
      const double pl = -g0 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-gC[s] + wt[s]) + w[s]; xn[s] = pk[s] - acc1[s] * dt; }
      publish<NB>(L, D, T, xn, lk - pl * dt);
      stage_products<NB, false, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs);
      const double cl = -g1 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) pk[s] = pk[s] - h * (acc1[s] + ((-gN[s] + wt[s]) + w[s])) + js[s];
      lnew = lk - h * (pl + cl) + jm;
    } else if (METHOD == VGPA_ODE_RK2) {
      stage_products<NB, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs);
      const double pl = -g0 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) xn[s] = pk[s] - h * ((-gC[s] + wt[s]) + w[s]);
      publish<NB>(L, D, T, xn, lk - h * pl);
      stage_products<NB, false, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs);
      const double cl = -(0.5 * (g1 + g0)) + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) pk[s] = pk[s] - dt * ((-(0.5 * (gN[s] + gC[s])) + wt[s]) + w[s]) + js[s];
      lnew = lk - dt * cl + jm;
    } else {  // RK4
      const double gmid = 0.5 * (g1 + g0);
      stage_products<NB, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs);
      const double k1 = -g0 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-gC[s] + wt[s]) + w[s]; xn[s] = pk[s] - h * acc1[s]; }
      publish<NB>(L, D, T, xn, lk - h * k1);
      stage_products<NB, false, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs);
      const double k2 = -gmid + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        acc2[s] = (-(0.5 * (gN[s] + gC[s])) + wt[s]) + w[s];
        xn[s] = pk[s] - h * acc2[s];
      }
      publish<NB>(L, D, T, xn, lk - h * k2);
      stage_products<NB, false, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs);
      const double k3 = -gmid + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        r[s] = (-(0.5 * (gN[s] + gC[s])) + wt[s]) + w[s];
        acc2[s] = acc2[s] + r[s];
        xn[s] = pk[s] - dt * r[s];
      }
      publish<NB>(L, D, T, xn, lk - dt * k3);
      stage_products<NB, false, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs);
      const double k4 = -g1 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        r[s] = (-gN[s] + wt[s]) + w[s];
        pk[s] = pk[s] - dt * (acc1[s] + 2.0 * acc2[s] + r[s]) / 6.0 + js[s];
      }
      lnew = lk - dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0 + jm;
    }
    lk = lnew;
    double* po = psi + (size_t)(t - 1) * DD;
#pragma unroll
    for (int s = 0; s < MAXU; s++) {
      if ((T.valid >> s) & 1u) po[T.gofs[s]] = pk[s];
      gC[s] = gN[s];
      gN[s] = (((T.valid >> s) & 1u) && t >= 2) ? gs[(size_t)(t - 2) * DD + T.gofs[s]] : 0.0;
    }
    if (vlane) lam[(size_t)(t - 1) * D + lane] = lk;
    publish<NB>(L, D, T, pk, lk);
    double* tmp = L.A0; L.A0 = L.A1; L.A1 = tmp;
  }
}

template <int METHOD, bool FWD, int NB>
hipError_t launch_nb(const OdeArgs& a, hipStream_t st) {
  constexpr size_t lds = Geo<NB>::LDS_DOUBLES * sizeof(double);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = FWD ? k_fwd_mfma<METHOD, NB> : k_bwd_mfma<METHOD, NB>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(a.batch), dim3(NT), lds, st, a);
  return hipGetLastError();
}

}  // namespace mfma

// One instantiation set per stepper (defined in ode_mfma_m<METHOD>.hip).
template <int METHOD> bool mfma_method_supported(int nb);
template <int METHOD> hipError_t mfma_method_launch(bool fwd, const OdeArgs& a, hipStream_t st);

}  // namespace vgpa
