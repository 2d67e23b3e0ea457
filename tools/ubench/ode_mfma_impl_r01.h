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

// A problem is integrated by ONE workgroup of NW waves: NW = 4 puts one wave on each SIMD of the CU; NW = 8 puts two,
// so that while one wave of a SIMD sits in the LDS exchange / element-wise part of a stage the other one can issue
// MFMAs (a lone wave per SIMD has nothing to hide its LDS and barrier latencies behind), and each lane owns half as
// many matrix elements, which keeps the kernel inside the 256-register budget of two waves per SIMD.
__device__ __forceinline__ int ltid() { return threadIdx.x; }
__device__ __forceinline__ int lwave() { return threadIdx.x >> 6; }

// Diagnostic build only (tools/ubench/ode_stamp.hip): per-segment cycle sums of one wave.  Never defined in the
// product build, so no stamp executes there.
#ifdef VGPA_STAMPS
__device__ long long g_stamp[8][16];
__device__ long long g_clk[4];   // s_memtime / s_memrealtime at kernel start and end (block 0, thread 0)
#define VGPA_STAMP(i)                                                                         \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    const long long t_ = clock64();                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_stamp[threadIdx.x >> 6][i] += t_ - stamp_prev_; \
    stamp_prev_ = clock64();                                                                  \
  } while (0)
#define VGPA_STAMP_DECL long long stamp_prev_ = clock64()
#define VGPA_STAMP_ARG , long long& stamp_prev_
#define VGPA_STAMP_PASS , stamp_prev_
#else
#define VGPA_STAMP(i) do {} while (0)
#define VGPA_STAMP_DECL do {} while (0)
#define VGPA_STAMP_ARG
#define VGPA_STAMP_PASS
#endif
#ifdef VGPA_WPE            // ubench only: force the register budget of VGPA_WPE waves per SIMD
#define VGPA_OCC __attribute__((amdgpu_waves_per_eu(VGPA_WPE, VGPA_WPE)))
#else
#define VGPA_OCC
#endif
constexpr int kMaxNB = 11;   // D <= 44: beyond that the backward kernel spills registers (generic path instead)

// ---- dealing units to (wave, slot) -------------------------------------------------------------------------
// Units are numbered group-major: NQ full column groups of NB units each, then NLEFT left-over units.  All units of
// a group share one B fragment.  Every wave loads TWO B fragments per k-step and slot s uses the first one when
// s < S1 and the second one otherwise -- a compile-time choice (a per-slot run-time select between MFMAs costs
// ~20 cycles per MFMA, measured).  The greedy below gives each wave units of at most two groups such that each
// group fits one of the two slot ranges; a wave fed by one group uses both ranges for it.
struct WaveDeal { int uA, nA, gA, uB, nB, gB; };   // first unit / count / group of the A-range and of the B-range

__host__ __device__ constexpr int deal_group_size(int nb, int nq, int nleft, int g) { return g < nq ? nb : (g == nq ? nleft : 0); }

// Returns the deal of wave `want` (0..nw-1); *done = all units were dealt.  Every wave aims at an even share of what is
// left (ceil(remaining units / remaining waves), at most maxu), so the matrix-core work of the SIMDs is balanced.
__host__ __device__ constexpr WaveDeal deal_units(int nb, int nq, int nleft, int maxu, int nw, int want, bool* done) {
  const int ngroups = nq + (nleft ? 1 : 0);
  const int s1 = maxu / 2, s2 = maxu - s1;
  int g = 0, off = 0;
  int left = nb * nq + nleft;
  WaveDeal res{0, 0, 0, 0, 0, 0};
  for (int w = 0; w < nw; w++) {
    WaveDeal d{0, 0, 0, 0, 0, 0};
    int quota = (left + (nw - w) - 1) / (nw - w);
    quota = quota < maxu ? quota : maxu;
    while (g < ngroups && off >= deal_group_size(nb, nq, nleft, g)) { g++; off = 0; }
    if (g < ngroups && quota > 0) {
      const int rem = deal_group_size(nb, nq, nleft, g) - off;
      const int a = rem < quota ? rem : quota;
      if (a <= s1 || a <= s2) {
        const bool first_in_a = a <= s1;
        const int u_first = g * nb + off, g_first = g;
        off += a;
        while (g < ngroups && off >= deal_group_size(nb, nq, nleft, g)) { g++; off = 0; }
        int b = 0, u_second = 0, g_second = g_first;
        if (g < ngroups && a < quota) {
          int cap = first_in_a ? s2 : s1;
          cap = cap < quota - a ? cap : quota - a;
          const int rem2 = deal_group_size(nb, nq, nleft, g) - off;
          b = rem2 < cap ? rem2 : cap;
          u_second = g * nb + off; g_second = g;
          off += b;
        }
        if (first_in_a) d = WaveDeal{u_first, a, g_first, u_second, b, g_second};
        else d = WaveDeal{u_second, b, g_second, u_first, a, g_first};
      } else {
        const int na = a < s1 ? a : s1;
        d = WaveDeal{g * nb + off, na, g, g * nb + off + na, a - na, g};
        off += a;
      }
    }
    left -= d.nA + d.nB;
    if (w == want) res = d;
  }
  if (done) *done = (left == 0);
  return res;
}

__host__ __device__ constexpr bool deal_fits(int nb, int nq, int nleft, int maxu, int nw) {
  bool ok = false;
  (void)deal_units(nb, nq, nleft, maxu, nw, 0, &ok);
  return ok;
}

__host__ __device__ constexpr int deal_min_slots(int nb, int nq, int nleft, int nu, int nw) {
  int m = (nu + nw - 1) / nw;
  while (!deal_fits(nb, nq, nleft, m, nw)) m++;
  return m;
}

// Compile-time geometry of the padded problem: NB = ceil(D/4) 4x4 blocks per dimension.
template <int NB_, int NW_>
struct Geo {
  static constexpr int NB = NB_;
  static constexpr int NW = NW_;                           // waves per problem
  static constexpr int NT = 64 * NW_;                      // threads per problem
  static constexpr int NQ = NB / 4;                       // full 16-column groups
  static constexpr int REM = NB % 4;                      // left-over column blocks per block-row
  static constexpr int G = REM ? 4 / REM : 0;             // block-rows packed into one left-over unit
  static constexpr int NLEFT = REM ? (NB + G - 1) / G : 0;
  static constexpr int NU = NB * NQ + NLEFT;              // units (MFMA accumulators) per product
  static constexpr int MAXU = deal_min_slots(NB, NQ, NLEFT, NU, NW);   // unit slots per wave (>= ceil(NU/NW))
  static constexpr int S1 = MAXU / 2;                     // slots [0,S1) use B fragment 0, slots [S1,MAXU) fragment 1
  static constexpr int P = 4 * NB;                        // padded dimension
  // LDS operand layouts ("k-pair interleaved"): element (k, c) of an operand matrix sits at
  //     ((k >> 3) * 4 + ((k >> 1) & 3)) * LD + 2 * c + (k & 1)
  // Rows 2j and 2j+1 share a 16-byte unit.  The eight rows 8g .. 8g+7 feed TWO k-steps of the 4x4x4 instruction: lane
  // group r4 supplies row 8g + 2 r4 to the first and row 8g + 2 r4 + 1 to the second (any split of the eight rows into
  // two sets of four works as long as both operands use the same one), so ONE ds_read_b128 per lane
  // fetches both -- and the lanes r4 = 0, 1 of a 32-lane store group, which own ADJACENT rows of the stage state, write
  // the two halves of the same units instead of colliding on the same banks (a lone wave per SIMD reaches the LDS rate with b128 reads but only ~1/5 of it with b64 reads,
  // MI355X_MICROARCH.md s.LDS).  LDX = 0 (mod 32) doubles makes the 16-wide B rows conflict-free for the b128 lane
  // groups; LDA: see below (the fragment reads themselves, 16-lane groups inside one operand row, do not depend on it).
  static constexpr int KKE = NB + (NB & 1);               // k-steps rounded up to even (extra rows are zero)
  static constexpr int ROWS = 2 * KKE;                    // (KKE / 2) k-pairs x 4 rows
  static constexpr int LDX = 32 * ((2 * P + 31) / 32);
  // LDA = 18 (mod 32) doubles, >= 2P: (i) 16 units one operand-row pair apart (the forward staging, which reads A in
  // whole rows) land on 32 different banks, (ii) so do the 32 rows a lane group of the backward mat-vec reads
  // (18 rho mod 32 runs over the even residues).  The fragment reads (b128, 16-lane groups inside one row) do not care.
  static constexpr int LDA = 32 * ((2 * P - 18 + 31) / 32) + 18;
  static constexpr int LDW = 32 * ((P + 31) / 32);        // exchange buffer for W^T (swizzled inside 32-column groups, w_off)
  static constexpr int EPT = (P * P / 2 + NT - 1) / NT;   // 16-byte operand units (two A entries) per thread for the HBM -> LDS staging
  static constexpr int TRASH = 2 * NT;                    // one 16-byte scratch slot per thread for lanes without an element
  static constexpr size_t LDS_DOUBLES = (size_t)ROWS * LDX + (size_t)P * LDW + 3 * (size_t)ROWS * LDA +
                                        (size_t)(2 + NW) * P + TRASH + 8;
};

// offset of element (k, c) in a k-pair interleaved operand matrix with leading dimension LD
__host__ __device__ constexpr int pair_off(int k, int c, int LD) { return ((k >> 3) * 4 + ((k >> 1) & 3)) * LD + 2 * c + (k & 1); }

// offset of W[r][c] in the exchange buffer.  A 32-lane store group writes rows R, R+1 (R even) x 16 consecutive columns
// and a load group reads 16 consecutive rows x columns R, R+1 (the transposed element of every lane): with bank =
// c + 16 (r & 1) + 2 (r >> 1) (mod 32) both patterns touch 32 different banks (a plain odd leading dimension leaves
// 7 two-way conflicts in every store).
__host__ __device__ constexpr int w_off(int r, int c, int LD) { return r * LD + (c & ~31) + ((c + 16 * (r & 1) + 2 * (r >> 1)) & 31); }

template <int NB, int NW>
struct Lds {
  double* X;     // [ROWS][LDX]  stage state (k-pair interleaved)
  double* W;     // [P][LDW]     exchange buffer for W^T
  double* A0;    // [ROWS][LDA]  operand of A at the step's start point
  double* AM;    // [ROWS][LDA]  operand of the mid-point
  double* A1;    // [ROWS][LDA]  operand of A at the step's end point
  double* xv;    // [P]          stage vector (m or lam)
  double* pv;    // [NW][P]      partial mat-vec sums
  double* trash; // [NT]         write/read target of lanes that own no matrix element in a slot
  __device__ __forceinline__ void carve(double* smem) {
    using g = Geo<NB, NW>;
    X = smem; W = X + g::ROWS * g::LDX; A0 = W + g::P * g::LDW; AM = A0 + g::ROWS * g::LDA;
    A1 = AM + g::ROWS * g::LDA; xv = A1 + g::ROWS * g::LDA; pv = xv + 2 * g::P; trash = pv + NW * g::P;
  }
};

template <int NB, int NW>
struct Tab {
  static constexpr int MAXU = Geo<NB, NW>::MAXU;
  int colA[MAXU];   // 2*(4*I_b + (l&3))
  int colB0, colB1; // B-fragment columns of the group feeding slots [0,S1) / slots [S1,MAXU)
  int offWw[MAXU];  // w_off(row, col)
  int offWr[MAXU];  // w_off(col, row)
  int offX[MAXU];   // pair_off(row, col, LDX)
  int gofs[MAXU];   // row*D + col   (global element offset inside a D x D matrix)
  unsigned valid;   // per-lane bit s: this lane owns a real matrix element in slot s
};

template <int NB, int NW>
__device__ __forceinline__ void build_tab(int D, Tab<NB, NW>& T) {
  using g = Geo<NB, NW>;
  const int lane = threadIdx.x & 63, wave = lwave();
  const int b = (lane >> 2) & 3, r4 = lane >> 4, c4 = lane & 3;
  const WaveDeal deal = deal_units(g::NB, g::NQ, g::NLEFT, g::MAXU, NW, wave, nullptr);
  constexpr int rem = g::REM ? g::REM : 1;
  auto group_col = [&](int grp) { return (grp < g::NQ) ? (16 * grp + (lane & 15)) : (4 * (4 * g::NQ + b % rem) + c4); };
  T.colB0 = 2 * group_col(deal.gA);      // (doubles; the pair layout stores two k-steps per column)
  T.colB1 = 2 * group_col(deal.gB);
  T.valid = 0u;
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) {
    const bool in_a = s < g::S1;
    const int t = in_a ? s : s - g::S1;
    const bool have = in_a ? (t < deal.nA) : (t < deal.nB);
    const int u = (in_a ? deal.uA : deal.uB) + t;
    int Ib = 0, Jb = 0;
    bool ok = false;
    if (have) {
      if (u < g::NB * g::NQ) {
        const int q = u / g::NB;
        Ib = u - q * g::NB; Jb = 4 * q + b; ok = true;
      } else {
        const int v = u - g::NB * g::NQ;
        const int i0 = v * g::G;
        Ib = i0 + b / rem; Jb = 4 * g::NQ + b % rem;
        ok = (b < g::G * g::REM) && (Ib < g::NB);
        if (!ok) { Ib = i0; Jb = 4 * g::NQ; }
      }
    }
    const int row = 4 * Ib + r4, col = 4 * Jb + c4;
    const bool own = ok && row < D && col < D;
    // offsets are relative to the LDS base; lanes without an element are pointed at their private trash word
    constexpr int W_BASE = g::ROWS * g::LDX;
    constexpr int TRASH_BASE = g::ROWS * g::LDX + g::P * g::LDW + 3 * g::ROWS * g::LDA + (2 + NW) * g::P;
    T.colA[s] = 2 * (4 * Ib + c4);
    T.offWw[s] = own ? (W_BASE + w_off(row, col, g::LDW)) : (TRASH_BASE + ltid());
    T.offWr[s] = own ? (W_BASE + w_off(col, row, g::LDW)) : (TRASH_BASE + ltid());
    T.offX[s] = own ? pair_off(row, col, g::LDX) : (TRASH_BASE + ltid());
    T.gofs[s] = row * D + col;
    if (own) T.valid |= (1u << s);
  }
}

// ---- one D^3 product on the matrix cores: w[s] = sum_kk Aop-block x X-block ---------------------------------
// LDAOP = leading dimension of the A-operand matrix (LDA, or LDX when the stage state itself is the operand).
// Straight-line code: KKE k-steps, fragments of step kk+1 are loaded while the MFMAs of step kk issue.
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int NB, int NW, int LDAOP>
__device__ __forceinline__ void mfma_product(const double* __restrict__ Aop, const double* __restrict__ X,
                                             const Tab<NB, NW>& T, double (&w)[Geo<NB, NW>::MAXU]) {
  using g = Geo<NB, NW>;
  constexpr int MAXU = g::MAXU, NP = g::KKE / 2;
  const int r4 = (threadIdx.x & 63) >> 4;
  const double* pa = Aop + r4 * LDAOP;
  const double* px = X + r4 * g::LDX;
  d2_t af[2][MAXU], bf[2][2];
#pragma unroll
  for (int s = 0; s < MAXU; s++) { w[s] = 0.0; af[0][s] = *reinterpret_cast<const d2_t*>(pa + T.colA[s]); }
  bf[0][0] = *reinterpret_cast<const d2_t*>(px + T.colB0);
  bf[0][1] = *reinterpret_cast<const d2_t*>(px + T.colB1);
#pragma unroll
  for (int kp = 0; kp < NP; kp++) {
    const int cur = kp & 1, nxt = cur ^ 1;
    if (kp + 1 < NP) {
#pragma unroll
      for (int s = 0; s < MAXU; s++) af[nxt][s] = *reinterpret_cast<const d2_t*>(pa + (kp + 1) * 4 * LDAOP + T.colA[s]);
      bf[nxt][0] = *reinterpret_cast<const d2_t*>(px + (kp + 1) * 4 * g::LDX + T.colB0);
      bf[nxt][1] = *reinterpret_cast<const d2_t*>(px + (kp + 1) * 4 * g::LDX + T.colB1);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        const double b = (s < g::S1) ? bf[cur][0][h] : bf[cur][1][h];   // compile-time choice
#if defined(VGPA_ABL_NOMFMA)
        w[s] += af[cur][s][h] * 1e-300 + b * 1e-300;                    // timing-only ablation (wrong results)
#else
        w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[cur][s][h], b, w[s], 0, 0, 0);
#endif
      }
    }
  }
}

// partial mat-vec of this wave: the NW waves split the padded k range [0, 4*KKE) into NW equal pieces (padding rows
// of the operand and padding entries of xv are zero): forward sum_k Aop[k][i] v[k], backward sum_k Aop[i][k] v[k].
// Branch-free.
template <int NB, int NW, bool FWD>
__device__ __forceinline__ double matvec_partial(const double* __restrict__ Aop, const double* __restrict__ xv) {
  using g = Geo<NB, NW>;
  constexpr int KQ = (4 * g::KKE) / NW;
  static_assert(KQ * NW == 4 * g::KKE, "the waves split the padded k range evenly");
  const int lane = threadIdx.x & 63, wave = lwave();
  const int k0 = wave * KQ;
  const int li = (lane < g::P) ? lane : 0;
  double av[KQ], xk[KQ];
#pragma unroll
  for (int k = 0; k < KQ; k++) {
    av[k] = FWD ? Aop[pair_off(k0 + k, li, g::LDA)] : Aop[pair_off(li, k0 + k, g::LDA)];
    xk[k] = xv[k0 + k];
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < KQ; k++) s = __builtin_fma(av[k], xk[k], s);
  return s;
}

// Products of one stage + the LDS exchange.  On return: w = own W element, wt = W^T element, vsum = (Aop-matvec)
// for lanes < D of wave 0.  Contains ONE barrier.
template <int NB, int NW, bool FWD, int LDAOP>
__device__ __forceinline__ void stage_products(const Lds<NB, NW>& L, int D, const double* Aop, const Tab<NB, NW>& T,
                                               const double* Avec, double (&w)[Geo<NB, NW>::MAXU],
                                               double (&wt)[Geo<NB, NW>::MAXU], double& vsum VGPA_STAMP_ARG) {
  using g = Geo<NB, NW>;
  const int lane = threadIdx.x & 63, wave = lwave();
  VGPA_STAMP(0);                       // elementwise work since the last publish
  mfma_product<NB, NW, LDAOP>(Aop, L.X, T, w);
  VGPA_STAMP(1);                       // MFMA product
#if defined(VGPA_ABL_NOMATVEC)
  const double part = 0.0;                                          // timing-only ablation (wrong results)
#else
  const double part = matvec_partial<NB, NW, FWD>(Avec, L.xv);
#endif
  VGPA_STAMP(2);                       // mat-vec
#if defined(VGPA_ABL_NOXCHG)
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) wt[s] = w[s];                    // timing-only ablation (wrong results)
  vsum = part;
  return;
#endif
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) L.X[T.offWw[s]] = w[s];          // (offsets are LDS-base relative; X is the base)
  L.pv[wave * g::P + ((lane < g::P) ? lane : 0)] = part;            // lanes >= P hold the same value as lane 0
  VGPA_STAMP(3);                       // W / pv stores
  __syncthreads();
  VGPA_STAMP(4);                       // barrier A
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) wt[s] = L.X[T.offWr[s]];
  vsum = 0.0;
  if (wave == 0 && lane < D) {
    vsum = L.pv[lane];
#pragma unroll
    for (int q = 1; q < NW; q++) vsum += L.pv[q * g::P + lane];
  }
  VGPA_STAMP(5);                       // W^T / pv loads
}

// publish the next stage state (matrix elements owned by this lane + vector entries of wave 0).  ONE barrier.
template <int NB, int NW>
__device__ __forceinline__ void publish(const Lds<NB, NW>& L, int D, const Tab<NB, NW>& T,
                                        const double (&xn)[Geo<NB, NW>::MAXU], double vn VGPA_STAMP_ARG) {
  VGPA_STAMP(6);                       // elementwise work of the stage
#pragma unroll
  for (int s = 0; s < Geo<NB, NW>::MAXU; s++) L.X[T.offX[s]] = xn[s];
  if (lwave() == 0 && (threadIdx.x & 63) < D) L.xv[threadIdx.x & 63] = vn;
  VGPA_STAMP(7);                       // X stores
#if !defined(VGPA_ABL_NOBARB)
  __syncthreads();
#endif
  VGPA_STAMP(8);                       // barrier B
}

// Values loaded from HBM before the time loop and only read inside it (Sigma, the constant jump, ...): make the compiler
// wait for them HERE.  Otherwise its wait-count pass, which cannot see across the loop back-edge that they arrived long
// ago, puts an s_waitcnt vmcnt(0) in front of their first use inside the loop -- and that also waits for every prefetch
// the step has just issued, i.e. it exposes a full memory latency per step (measured on the backward kernel).
__device__ __forceinline__ void settle(double& v) { asm volatile("" : "+v"(v)); }

// ---- A(t): HBM -> registers -> LDS operand buffer, in 16-byte operand units ----------------------------------------
// The k-pair interleaved operand layout keeps rows 2p and 2p+1 of the operand in one 16-byte unit per column.  A staging
// item is such a unit: (p, o) = rows 2p, 2p+1 of the operand at column o -- forward (operand = A^T) the elements
// A[o][2p], A[o][2p+1], backward (operand = A) A[2p][o], A[2p+1][o].  Items are dealt with o fastest over the threads,
// so a 16-lane group writes 16 consecutive units with ONE ds_write_b128: conflict-free.  (The first version wrote single
// elements with the matrix column fastest: forward that is a transposing store whose 32-lane groups hit 8 banks, a
// four-way conflict on every one of the 64 wave-stores of a step -- ~2 k LDS cycles per step.)  The forward HBM reads
// become strided 8-byte loads (one row per lane); they are prefetched a whole step ahead and every 128-byte line is
// still fetched once (the other lanes of the same instruction group use the rest of it).
typedef double a2_t __attribute__((ext_vector_type(2)));

template <int NB, int NW>
struct AStage {
  int g0[Geo<NB, NW>::EPT];   // global element offset of the unit's first entry (-1: none)
  int g1[Geo<NB, NW>::EPT];   // ... of its second entry (-1: padding row)
  int lo[Geo<NB, NW>::EPT];   // LDS offset of the unit inside an operand buffer (-1: no item)
};

template <int NB, int NW, bool FWD>
__device__ __forceinline__ void build_astage(int D, AStage<NB, NW>& s) {
  using g = Geo<NB, NW>;
  const int npair = (D + 1) / 2;
#pragma unroll
  for (int q = 0; q < g::EPT; q++) {
    const int e = ltid() + q * g::NT;
    // backward: column o fastest over the lanes (rows of A are contiguous in o); forward: pair p fastest (the two
    // entries A[o][2p], A[o][2p+1] of consecutive p are contiguous: whole rows of A per 20 lanes)
    const int p = FWD ? e % npair : e / D, o = FWD ? e / npair : e - (e / D) * D;
    const bool ok = FWD ? (o < D) : (p < npair);
    const bool two = ok && (2 * p + 1 < D);
    s.g0[q] = ok ? (FWD ? o * D + 2 * p : 2 * p * D + o) : 0;
    s.g1[q] = two ? (FWD ? o * D + 2 * p + 1 : (2 * p + 1) * D + o) : 0;
    s.lo[q] = ok ? pair_off(2 * p, o, g::LDA) : -1;
  }
}

// Branch-free: lanes without an item load element 0 (always valid) and store into the workgroup's trash area -- a
// conditional load / store costs an EXEC-masked branch each, and there are 6 EPT of them per step.
template <int NB, int NW>
__device__ __forceinline__ void load_a(const double* __restrict__ A, const AStage<NB, NW>& s, a2_t (&a)[Geo<NB, NW>::EPT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB, NW>::EPT; q++) {
    a[q][0] = A[s.g0[q]];      // (offsets of lanes without an item are clamped to 0 in build_astage; their value is
    a[q][1] = A[s.g1[q]];      //  never stored, or -- second entry of the last pair when D is odd -- hits a row that only meets zeros)
  }
}

// registers -> LDS operand buffer (MID: the mid-point 0.5 * (a0 + a1)); `trash16` = this thread's 16-byte trash slot
template <int NB, int NW, bool MID>
__device__ __forceinline__ void store_a(double* __restrict__ buf, double* __restrict__ trash16, const AStage<NB, NW>& s,
                                        const a2_t (&a0)[Geo<NB, NW>::EPT], const a2_t (&a1)[Geo<NB, NW>::EPT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB, NW>::EPT; q++) {
    a2_t v = a0[q];
    if (MID) { v[0] = 0.5 * (a0[q][0] + a1[q][0]); v[1] = 0.5 * (a0[q][1] + a1[q][1]); }
    double* dst = (s.lo[q] >= 0) ? buf + s.lo[q] : trash16;
    *reinterpret_cast<a2_t*>(dst) = v;
  }
}

// =================================================================================================================
template <int METHOD, int NB, int NW>
__global__ void __launch_bounds__(64 * NW) VGPA_OCC k_fwd_mfma(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = Geo<NB, NW>;
  constexpr int MAXU = g::MAXU, EPT = g::EPT;
  const int D = a.D, DD = D * D, Np = a.Np;
  constexpr int NT = g::NT;
  const int prob = (int)blockIdx.x;
  const int tid = ltid(), lane = tid & 63, wave = lwave();
  double* lds_base = smem;
  Lds<NB, NW> L;
  L.carve(lds_base);
  const double* A = a.A + (size_t)prob * a.strideA;
  const double* bb = a.b + (size_t)prob * a.strideB;
  double* mt = a.m + (size_t)prob * Np * D;
  double* st = a.S + (size_t)prob * Np * DD;
  const double dt = a.dt, h = 0.5 * a.dt;
  const bool vlane = (wave == 0) && (lane < D);

  VGPA_STAMP_DECL;
#ifdef VGPA_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0) { g_clk[0] = __builtin_amdgcn_s_memtime(); g_clk[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
  Tab<NB, NW> T;
  build_tab<NB, NW>(D, T);
  AStage<NB, NW> AS;
  build_astage<NB, NW, true>(D, AS);
  double* trash16 = L.trash + 2 * tid;
  for (int i = tid; i < (int)g::LDS_DOUBLES; i += NT) lds_base[i] = 0.0;
  __syncthreads();

  double sk[MAXU], sig[MAXU], w[MAXU], wt[MAXU], r[MAXU], acc1[MAXU], acc2[MAXU], xn[MAXU];
  a2_t aC[EPT], aN[EPT];
  double mk = 0.0, vs = 0.0;
#pragma unroll
  for (int s = 0; s < MAXU; s++) {
    const bool ok = (T.valid >> s) & 1u;
    sk[s] = ok ? a.S0[T.gofs[s]] : 0.0;
    sig[s] = ok ? a.Sigma[T.gofs[s]] : 0.0;
    acc1[s] = acc2[s] = 0.0;
    if (ok) st[T.gofs[s]] = sk[s];
    L.X[T.offX[s]] = sk[s];
  }
  if (vlane) { mk = a.m0[lane]; mt[lane] = mk; L.xv[lane] = mk; }
  load_a<NB, NW>(A, AS, aC);
  store_a<NB, NW, false>(L.A0, trash16, AS, aC, aC);
  if (Np > 1) load_a<NB, NW>(A + DD, AS, aN);
  // offset vectors: b0 = b_k, b1 = b_{k+1}; b_{k+2} is fetched one step ahead (HBM latency off the critical path)
  double b0 = vlane ? bb[lane] : 0.0;
  double b1 = (vlane && Np > 1) ? bb[D + lane] : 0.0;
#pragma unroll
  for (int s = 0; s < MAXU; s++) { settle(sk[s]); settle(sig[s]); }
  settle(b0); settle(b1); settle(mk);
  __syncthreads();

  for (int k = 0; k < Np - 1; k++) {
    // S_k, m_k of the previous iteration go to HBM here, right behind the operand loads they follow in the memory
    // queue: by the time the next iteration waits for its operands (vmcnt) these stores have long retired.
#if !defined(VGPA_ABL_NOSTORE)
    if (k > 0) {
      double* so = st + (size_t)k * DD;
#pragma unroll
      for (int s = 0; s < MAXU; s++)
        if ((T.valid >> s) & 1u) so[T.gofs[s]] = sk[s];
      if (vlane) mt[(size_t)k * D + lane] = mk;
    }
#endif
    // operands of this step: A1 <- A_{k+1}, AM <- mid-point; prefetch A_{k+2} for the next step
#if !defined(VGPA_ABL_NOSTAGE)
    store_a<NB, NW, false>(L.A1, trash16, AS, aN, aN);
    if (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4) store_a<NB, NW, true>(L.AM, trash16, AS, aC, aN);
#pragma unroll
    for (int q = 0; q < EPT; q++) aC[q] = aN[q];
    if (k + 2 < Np) load_a<NB, NW>(A + (size_t)(k + 2) * DD, AS, aN);
#endif
    const double b2 = (vlane && k + 2 < Np) ? bb[(size_t)(k + 2) * D + lane] : 0.0;
    double mnew = 0.0;

    if (METHOD == VGPA_ODE_EULER) {
      stage_products<NB, NW, true, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
#pragma unroll
      for (int s = 0; s < MAXU; s++) sk[s] = sk[s] + ((-w[s] - wt[s]) + sig[s]) * dt;
      mnew = mk + (-vs + b0) * dt;
    } else if (METHOD == VGPA_ODE_HEUN) {
      stage_products<NB, NW, true, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
      const double pm = -vs + b0;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-w[s] - wt[s]) + sig[s]; xn[s] = sk[s] + acc1[s] * dt; }
      publish<NB, NW>(L, D, T, xn, mk + pm * dt VGPA_STAMP_PASS);
      stage_products<NB, NW, true, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs VGPA_STAMP_PASS);
      const double cm = -vs + b1;
#pragma unroll
      for (int s = 0; s < MAXU; s++) sk[s] = sk[s] + h * (acc1[s] + ((-w[s] - wt[s]) + sig[s]));
      mnew = mk + h * (pm + cm);
    } else if (METHOD == VGPA_ODE_RK2) {
      // covariance predictor: S_k stands in for A_k (Q2): operand = X itself (S symmetric); mean predictor: A_k
      stage_products<NB, NW, true, g::LDX>(L, D, L.X, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
      const double pm = -vs + b0;
#pragma unroll
      for (int s = 0; s < MAXU; s++) xn[s] = sk[s] + h * ((-w[s] - wt[s]) + sig[s]);
      publish<NB, NW>(L, D, T, xn, mk + h * pm VGPA_STAMP_PASS);
      stage_products<NB, NW, true, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs VGPA_STAMP_PASS);
      const double cm = -vs + 0.5 * (b0 + b1);
#pragma unroll
      for (int s = 0; s < MAXU; s++) sk[s] = sk[s] + dt * ((-w[s] - wt[s]) + sig[s]);
      mnew = mk + dt * cm;
    } else {  // RK4
      const double bmid = 0.5 * (b0 + b1);
      stage_products<NB, NW, true, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
      const double k1 = -vs + b0;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-w[s] - wt[s]) + sig[s]; xn[s] = sk[s] + h * acc1[s]; }
      publish<NB, NW>(L, D, T, xn, mk + h * k1 VGPA_STAMP_PASS);
      stage_products<NB, NW, true, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs VGPA_STAMP_PASS);
      const double k2 = -vs + bmid;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc2[s] = (-w[s] - wt[s]) + sig[s]; xn[s] = sk[s] + h * acc2[s]; }
      publish<NB, NW>(L, D, T, xn, mk + h * k2 VGPA_STAMP_PASS);
      stage_products<NB, NW, true, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs VGPA_STAMP_PASS);
      const double k3 = -vs + bmid;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { r[s] = (-w[s] - wt[s]) + sig[s]; acc2[s] = acc2[s] + r[s]; xn[s] = sk[s] + dt * r[s]; }
      publish<NB, NW>(L, D, T, xn, mk + dt * k3 VGPA_STAMP_PASS);
      stage_products<NB, NW, true, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs VGPA_STAMP_PASS);
      const double k4 = -vs + b1;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        r[s] = (-w[s] - wt[s]) + sig[s];
        sk[s] = sk[s] + dt * (acc1[s] + 2.0 * acc2[s] + r[s]) / 6.0;
      }
      mnew = mk + dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0;
    }
    mk = mnew;
    publish<NB, NW>(L, D, T, sk, mk VGPA_STAMP_PASS);
    // rotate operand buffers: A_{k+1} becomes the start-point operand of the next step
    double* tmp = L.A0; L.A0 = L.A1; L.A1 = tmp;
    b0 = b1; b1 = b2;
  }
  if (Np > 1) {
    double* so = st + (size_t)(Np - 1) * DD;
#pragma unroll
    for (int s = 0; s < MAXU; s++)
      if ((T.valid >> s) & 1u) so[T.gofs[s]] = sk[s];
    if (vlane) mt[(size_t)(Np - 1) * D + lane] = mk;
  }
#ifdef VGPA_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0) { g_clk[2] = __builtin_amdgcn_s_memtime(); g_clk[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// =================================================================================================================
// DENSEJ: the jumps come as dense (Np, D, D) / (Np, D) arrays (operator-level API); otherwise one constant matrix jump
// applied at the observation indices and sparse vector jumps (the sweep).  A compile-time switch: with both paths in one
// kernel the dense path's loads and the sparse path's selects share registers, and the wait-count pass then puts an
// s_waitcnt vmcnt(0) in front of the selects -- behind the prefetches the step has just issued.
template <int METHOD, int NB, int NW, bool DENSEJ>
__global__ void __launch_bounds__(64 * NW) VGPA_OCC k_bwd_mfma(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = Geo<NB, NW>;
  constexpr int MAXU = g::MAXU, EPT = g::EPT;
  const int D = a.D, DD = D * D, Np = a.Np;
  constexpr int NT = g::NT;
  const int prob = (int)blockIdx.x;
  const int tid = ltid(), lane = tid & 63, wave = lwave();
  double* lds_base = smem;
  Lds<NB, NW> L;
  L.carve(lds_base);
  const double* A = a.A + (size_t)prob * a.strideA;
  const double* gm = a.dEm + (size_t)prob * Np * D;
  const double* gs = a.dEs + (size_t)prob * Np * DD;
  double* lam = a.lam + (size_t)prob * Np * D;
  double* psi = a.psi + (size_t)prob * Np * DD;
  const double dt = a.dt, h = 0.5 * a.dt;
  const bool vlane = (wave == 0) && (lane < D);

  VGPA_STAMP_DECL;
  Tab<NB, NW> T;
  build_tab<NB, NW>(D, T);
  AStage<NB, NW> AS;
  build_astage<NB, NW, false>(D, AS);
  double* trash16 = L.trash + 2 * tid;
  for (int i = tid; i < (int)g::LDS_DOUBLES; i += NT) lds_base[i] = 0.0;
  __syncthreads();

  double pk[MAXU], gC[MAXU], gN[MAXU], jsc[MAXU], w[MAXU], wt[MAXU], r[MAXU], acc1[MAXU], acc2[MAXU], xn[MAXU];
  a2_t aC[EPT], aN[EPT];
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
  load_a<NB, NW>(A + (size_t)(Np - 1) * DD, AS, aC);
  store_a<NB, NW, false>(L.A0, trash16, AS, aC, aC);
  if (Np > 1) load_a<NB, NW>(A + (size_t)(Np - 2) * DD, AS, aN);
  // per-step vectors are fetched one step ahead: g0 = dEsde_dm[t], g1 = dEsde_dm[t-1]; jump of index t-1
  double g0 = vlane ? gm[(size_t)(Np - 1) * D + lane] : 0.0;
  double g1 = (vlane && Np > 1) ? gm[(size_t)(Np - 2) * D + lane] : 0.0;
  // observation index of grid point t (-1: none): the one of t-1 decides this step's jump, the one of t-2 which vector
  // jump to prefetch; it is itself fetched a step before it is needed (its load must not be waited for in the step that
  // issues it: that wait would also cover the prefetches issued just before)
  const bool sparse = !DENSEJ && a.obs_idx;
  int n_obs_cur = (sparse && Np > 1) ? a.obs_idx[Np - 2] : -1;
  int n_obs_next = (sparse && Np > 2) ? a.obs_idx[Np - 3] : -1;
  // (the in-loop fetch goes through a lane-"dependent" address so that the value stays in a VGPR until the next step
  // reads it with v_readfirstlane; a visibly uniform load is moved to an SGPR -- i.e. waited for -- on the spot)
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  int n_obs_next2_v = -1;
  double jm = 0.0;
  if (Np > 1) {
    if (DENSEJ) { if (vlane) jm = a.jm_dense[((size_t)prob * Np + (Np - 2)) * D + lane]; }
    else if (vlane && n_obs_cur >= 0) jm = a.jm_sparse[((size_t)prob * a.n_obs + n_obs_cur) * D + lane];
  }
#pragma unroll
  for (int s = 0; s < MAXU; s++) { settle(jsc[s]); settle(gC[s]); settle(gN[s]); }
  settle(g0); settle(g1); settle(jm);
  __syncthreads();

  for (int t = Np - 1; t > 0; t--) {
    if (t < Np - 1) n_obs_next = __builtin_amdgcn_readfirstlane(n_obs_next2_v);   // fetched during the previous step
    // Psi_t, lam_t of the previous iteration go to HBM here (see the forward kernel)
    if (t < Np - 1) {
      double* po = psi + (size_t)t * DD;
#pragma unroll
      for (int s = 0; s < MAXU; s++)
        if ((T.valid >> s) & 1u) po[T.gofs[s]] = pk[s];
      if (vlane) lam[(size_t)t * D + lane] = lk;
    }
    store_a<NB, NW, false>(L.A1, trash16, AS, aN, aN);
    if (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4) store_a<NB, NW, true>(L.AM, trash16, AS, aN, aC);
#pragma unroll
    for (int q = 0; q < EPT; q++) aC[q] = aN[q];
    if (t >= 2) load_a<NB, NW>(A + (size_t)(t - 2) * DD, AS, aN);
    const double g2 = (vlane && t >= 2) ? gm[(size_t)(t - 2) * D + lane] : 0.0;   // for the next step
    n_obs_next2_v = (sparse && t >= 3) ? a.obs_idx[t - 3 + vzero] : -1;             // for the next step
    double jm_next = 0.0;
    if (t >= 2) {
      if (DENSEJ) { if (vlane) jm_next = a.jm_dense[((size_t)prob * Np + (t - 2)) * D + lane]; }
      else if (vlane && n_obs_next >= 0) jm_next = a.jm_sparse[((size_t)prob * a.n_obs + n_obs_next) * D + lane];
    }
    // matrix jump of index t-1
    double js[MAXU];
    if (DENSEJ) {
      const double* jp = a.js_dense + ((size_t)prob * Np + (t - 1)) * DD;
#pragma unroll
      for (int s = 0; s < MAXU; s++) js[s] = ((T.valid >> s) & 1u) ? jp[T.gofs[s]] : 0.0;
    } else {
#pragma unroll
      for (int s = 0; s < MAXU; s++) js[s] = (n_obs_cur >= 0) ? jsc[s] : 0.0;
    }
    double lnew = 0.0;

    if (METHOD == VGPA_ODE_EULER) {
      stage_products<NB, NW, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
#pragma unroll
      for (int s = 0; s < MAXU; s++) pk[s] = pk[s] - ((-gC[s] + wt[s]) + w[s]) * dt + js[s];
      lnew = lk - (-g0 + vs) * dt + jm;
    } else if (METHOD == VGPA_ODE_HEUN) {
      stage_products<NB, NW, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
      const double pl = -g0 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-gC[s] + wt[s]) + w[s]; xn[s] = pk[s] - acc1[s] * dt; }
      publish<NB, NW>(L, D, T, xn, lk - pl * dt VGPA_STAMP_PASS);
      stage_products<NB, NW, false, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs VGPA_STAMP_PASS);
      const double cl = -g1 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) pk[s] = pk[s] - h * (acc1[s] + ((-gN[s] + wt[s]) + w[s])) + js[s];
      lnew = lk - h * (pl + cl) + jm;
    } else if (METHOD == VGPA_ODE_RK2) {
      stage_products<NB, NW, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
      const double pl = -g0 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) xn[s] = pk[s] - h * ((-gC[s] + wt[s]) + w[s]);
      publish<NB, NW>(L, D, T, xn, lk - h * pl VGPA_STAMP_PASS);
      stage_products<NB, NW, false, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs VGPA_STAMP_PASS);
      const double cl = -(0.5 * (g1 + g0)) + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) pk[s] = pk[s] - dt * ((-(0.5 * (gN[s] + gC[s])) + wt[s]) + w[s]) + js[s];
      lnew = lk - dt * cl + jm;
    } else {  // RK4
      const double gmid = 0.5 * (g1 + g0);
      stage_products<NB, NW, false, g::LDA>(L, D, L.A0, T, L.A0, w, wt, vs VGPA_STAMP_PASS);
      const double k1 = -g0 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) { acc1[s] = (-gC[s] + wt[s]) + w[s]; xn[s] = pk[s] - h * acc1[s]; }
      publish<NB, NW>(L, D, T, xn, lk - h * k1 VGPA_STAMP_PASS);
      stage_products<NB, NW, false, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs VGPA_STAMP_PASS);
      const double k2 = -gmid + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        acc2[s] = (-(0.5 * (gN[s] + gC[s])) + wt[s]) + w[s];
        xn[s] = pk[s] - h * acc2[s];
      }
      publish<NB, NW>(L, D, T, xn, lk - h * k2 VGPA_STAMP_PASS);
      stage_products<NB, NW, false, g::LDA>(L, D, L.AM, T, L.AM, w, wt, vs VGPA_STAMP_PASS);
      const double k3 = -gmid + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        r[s] = (-(0.5 * (gN[s] + gC[s])) + wt[s]) + w[s];
        acc2[s] = acc2[s] + r[s];
        xn[s] = pk[s] - dt * r[s];
      }
      publish<NB, NW>(L, D, T, xn, lk - dt * k3 VGPA_STAMP_PASS);
      stage_products<NB, NW, false, g::LDA>(L, D, L.A1, T, L.A1, w, wt, vs VGPA_STAMP_PASS);
      const double k4 = -g1 + vs;
#pragma unroll
      for (int s = 0; s < MAXU; s++) {
        r[s] = (-gN[s] + wt[s]) + w[s];
        pk[s] = pk[s] - dt * (acc1[s] + 2.0 * acc2[s] + r[s]) / 6.0 + js[s];
      }
      lnew = lk - dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0 + jm;
    }
    lk = lnew;
#pragma unroll
    for (int s = 0; s < MAXU; s++) {
      gC[s] = gN[s];
      gN[s] = (((T.valid >> s) & 1u) && t >= 2) ? gs[(size_t)(t - 2) * DD + T.gofs[s]] : 0.0;
    }
    publish<NB, NW>(L, D, T, pk, lk VGPA_STAMP_PASS);
    double* tmp = L.A0; L.A0 = L.A1; L.A1 = tmp;
    g0 = g1; g1 = g2; jm = jm_next; n_obs_cur = n_obs_next;
  }
  if (Np > 1) {
    double* po = psi;
#pragma unroll
    for (int s = 0; s < MAXU; s++)
      if ((T.valid >> s) & 1u) po[T.gofs[s]] = pk[s];
    if (vlane) lam[lane] = lk;
  }
}

template <int METHOD, bool FWD, int NB, int NW>
hipError_t launch_nb_w(const OdeArgs& a, hipStream_t st) {
  constexpr size_t lds = Geo<NB, NW>::LDS_DOUBLES * sizeof(double);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = FWD ? k_fwd_mfma<METHOD, NB, NW> : (a.js_dense ? k_bwd_mfma<METHOD, NB, NW, true> : k_bwd_mfma<METHOD, NB, NW, false>);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(a.batch), dim3(64 * NW), lds, st, a);
  return hipGetLastError();
}

// Eight waves per problem once every wave still gets at least two MFMA units per k-step; four otherwise (or on request).
template <int METHOD, bool FWD, int NB>
hipError_t launch_nb(const OdeArgs& a, hipStream_t st) {
  if constexpr (Geo<NB, 8>::NU >= 16) {
    if (!a.four_waves) return launch_nb_w<METHOD, FWD, NB, 8>(a, st);
  }
  return launch_nb_w<METHOD, FWD, NB, 4>(a, st);
}

}  // namespace mfma

// One instantiation set per stepper (defined in ode_mfma_m<METHOD>.hip).
template <int METHOD> bool mfma_method_supported(int nb);
template <int METHOD> hipError_t mfma_method_launch(bool fwd, const OdeArgs& a, hipStream_t st);

}  // namespace vgpa
