// Symmetric fast path of the time-stepping kernels: fp64 matrix cores, role-specialised waves, two problems per workgroup.
// (Included by ode_mfma_m{0,1,2,3}.hip, one translation unit per stepper so that they compile in parallel.)
//
// Math.  With S (resp. Psi) symmetric the two products of the reference collapse to one:
//     forward : f_S   = -A S - S A^T + Sigma = -(W + W^T) + Sigma,   W  = A S        (ode_solver.py:60)
//     backward: f_Psi = -G + Psi A + A^T Psi = -G + W'^T + W',       W' = A^T Psi    (ode_solver.py:94)
// so a stage costs ONE D^3 product.  The steppers are those of src/numerics/{euler,heun,runge_kutta2,
// runge_kutta4}.py (incl. the RK2 covariance predictor that passes S_k as A, runge_kutta2.py:96, and
// f_lam = -g + A.lam, ode_solver.py:77); jumps are added after the step (euler.py:139-149).
//
// Mapping to gfx950 (round 2).  The recursion is a chain of dependent stages: product -> transpose exchange ->
// element-wise Runge-Kutta bookkeeping -> next product.  Round 1 ran that chain with every wave doing every part, all
// waves in lock step: ~3.2 k cycles per stage against 1.1 k cycles of MFMA issue, two workgroup barriers per stage, 208-244
// VGPRs of lane-owned RK state next to the MFMA fragments (one workgroup per CU).  Here the roles are split:
//   * P waves (4, one per SIMD) do nothing but the product on v_mfma_f64_4x4x4_4b_f64: fragments from LDS, W to LDS.
//     They own no state, so their registers go to fragment buffering.
//   * E waves (4 per problem, one per SIMD) own the RK state (S_k / Psi_t, the RK sums) as ROW-PAIR items
//     (rows 2p, 2p+1 at one column = one 16-byte unit of the k-pair interleaved operand layout): they read W[r][c] and
//     W[c][r] straight from the LDS copy the P waves left (the transpose is an address, not an exchange), apply the
//     stepper, publish the next stage state with one ds_write_b128 per item, do the mean / lambda recursion (mat-vec on
//     the VALU while the P waves are in the product), stage A(t) HBM -> registers -> LDS one step ahead and stream
//     S_k / Psi_t back to HBM.
//   * A workgroup integrates TWO problems: while the P waves multiply for problem A, the E waves of problem B do B's
//     element-wise stage, and vice versa.  One workgroup barrier per phase = per problem-stage (round 1: two), the matrix
//     pipe works in every phase, and the latency chain of one problem hides behind the product of the other.  With at
//     most one problem per CU (batch <= #CUs, e.g. the single problem of an SCG run) a workgroup takes one problem and the
//     phases alternate P / E.
// LDS per problem (D = 40): stage state X 15 KB + two A-operand buffers (start/end point R, mid-point M) 26 KB + W 13 KB
// + vectors; Sigma (or the constant matrix jump) once per workgroup; 131 KB for two problems.
#pragma once
#include "vgpa_internal.h"

namespace vgpa {
namespace mfma {

constexpr int kMaxNB = 11;       // D <= 44
constexpr int kMaxPairNB = 10;   // two problems per workgroup fit the 160 KB of LDS up to D = 40
constexpr int kNPW = 4;          // P waves (one per SIMD)
constexpr int kNE = 256;         // E threads per problem (4 waves, one per SIMD)

typedef double d2_t __attribute__((ext_vector_type(2)));

// ---- dealing MFMA units to (P wave, slot) -----------------------------------------------------------------------
// A "unit" is one MFMA accumulator = four 4x4 output blocks: (I, J = 4q..4q+3) for the full 16-column groups, and the
// left-over column blocks of several block-rows packed together (D = 40: 100 blocks = 25 units = 250 MFMAs per product).
// Units are numbered group-major: NQ full column groups of NB units each, then NLEFT left-over units.  All units of a
// group share one B fragment.  Every wave loads TWO B fragments per k-step and slot s uses the first one when s < S1 and
// the second one otherwise -- a compile-time choice.  The greedy below gives each wave units of at most two groups.
struct WaveDeal { int uA, nA, gA, uB, nB, gB; };   // first unit / count / group of the A-range and of the B-range

__host__ __device__ constexpr int deal_group_size(int nb, int nq, int nleft, int g) { return g < nq ? nb : (g == nq ? nleft : 0); }

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

__host__ __device__ constexpr int cmin(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }
__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// Compile-time geometry of the padded problem: NB = ceil(D/4) 4x4 blocks per dimension.
template <int NB_>
struct Geo {
  static constexpr int NB = NB_;
  static constexpr int NQ = NB / 4;                       // full 16-column groups
  static constexpr int REM = NB % 4;                      // left-over column blocks per block-row
  static constexpr int G = REM ? 4 / REM : 0;             // block-rows packed into one left-over unit
  static constexpr int NLEFT = REM ? (NB + G - 1) / G : 0;
  static constexpr int NU = NB * NQ + NLEFT;              // units (MFMA accumulators) per product
  static constexpr int MAXU = deal_min_slots(NB, NQ, NLEFT, NU, kNPW);   // unit slots per P wave
  static constexpr int S1 = MAXU / 2;                     // slots [0,S1) use B fragment 0, slots [S1,MAXU) fragment 1
  static constexpr int P = 4 * NB;                        // padded dimension
  // LDS operand layout ("k-pair interleaved"): element (k, c) of an operand matrix sits at (k >> 1) * LD + 2 c + (k & 1):
  // rows 2p and 2p+1 share a 16-byte unit.  Four consecutive row pairs feed TWO k-steps of the 4x4x4 instruction (lane
  // group r4 supplies row 8g + 2 r4 to the first and row 8g + 2 r4 + 1 to the second), so ONE ds_read_b128 per lane
  // fetches both.  LDX = 0 (mod 32) doubles makes the 16-wide B rows conflict-free for the b128 lane groups; LDA = 18
  // (mod 32): conflict-free for the staging stores (8 lanes one row pair apart) and the row-pair reads of the backward
  // mat-vec.
  static constexpr int KKE = NB + (NB & 1);               // k-steps rounded up to even (extra rows are zero)
  static constexpr int RP = 2 * KKE;                      // row pairs of an operand buffer
  static constexpr int NKP = KKE / 2;                     // k-pairs (fragment reads) per product
  static constexpr int LDX = 32 * ((2 * P + 31) / 32);
  static constexpr int LDA = 32 * ((2 * P - 18 + 31) / 32) + 18;
  // W[r][c] row-major with LDW = 2 (mod 4) doubles: the transposed 16-byte read (W[c][2p], W[c][2p+1]) of consecutive c
  // touches all 64 banks once per 16 lanes; rows are 16-byte aligned.
  static constexpr int LDW = P + 2;
  static constexpr int NIT = cdiv((P / 2) * P, kNE);      // row-pair items per E thread
  // mat-vec partial sums on the E threads: forward lane = (column i, part of the row pairs), backward lane = (row pair,
  // part of the columns)
  static constexpr int NPARTF = cmin(kNE / P, RP);
  static constexpr int RPP = cdiv(RP, NPARTF);
  static constexpr int NPARTB = cmin(kNE / (P / 2), P);
  static constexpr int KPP = cdiv(P, NPARTB);
  // per-problem LDS regions (doubles)
  static constexpr int XS = RP * LDX;
  static constexpr int AS = RP * LDA;
  static constexpr int WS = P * LDW + 64 * kNPW;          // + one trash double per P lane
  static constexpr int XV = 4 * KKE + 4;                  // stage vector, padded like the operand rows
  static constexpr int PV = cmax(NPARTF, NPARTB) * P;
  static constexpr int PROB = XS + 2 * AS + WS + XV + PV;
  static constexpr int SIGS = P * P;                      // Sigma / constant matrix jump in item layout
  static constexpr size_t lds_doubles(int nprob) { return (size_t)nprob * PROB + SIGS + 2 * kNE; }
};

template <int NB>
struct Lds {
  double* X;     // [RP][LDX]  stage state (k-pair interleaved)
  double* R;     // [RP][LDA]  A-operand at the start point of the step, then at its end point
  double* M;     // [RP][LDA]  A-operand of the mid-point
  double* W;     // [P][LDW]   product, row-major (+ P-lane trash)
  double* xv;    // [4 KKE]    stage vector (m or lam)
  double* pv;    // [NPART][P] partial mat-vec sums
  __device__ __forceinline__ void carve(double* base) {
    using g = Geo<NB>;
    X = base; R = X + g::XS; M = R + g::AS; W = M + g::AS; xv = W + g::WS; pv = xv + g::XV;
  }
};

// ---- P waves -----------------------------------------------------------------------------------------------------
template <int NB>
struct PTab {
  static constexpr int MAXU = Geo<NB>::MAXU;
  int colA[MAXU];   // 2*(4*I_b + (l&3))
  int colB0, colB1; // B-fragment columns of the group feeding slots [0,S1) / slots [S1,MAXU)
  int offW[MAXU];   // row * LDW + col of the element this lane's accumulator holds (trash for lanes without one)
};

template <int NB>
__device__ __forceinline__ void build_ptab(int pw, int lane, PTab<NB>& T) {
  using g = Geo<NB>;
  const int b = (lane >> 2) & 3, r4 = lane >> 4, c4 = lane & 3;
  const WaveDeal deal = deal_units(g::NB, g::NQ, g::NLEFT, g::MAXU, kNPW, pw, nullptr);
  constexpr int rem = g::REM ? g::REM : 1;
  auto group_col = [&](int grp) { return (grp < g::NQ) ? (16 * grp + (lane & 15)) : (4 * (4 * g::NQ + b % rem) + c4); };
  T.colB0 = 2 * group_col(deal.gA);
  T.colB1 = 2 * group_col(deal.gB);
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
    T.colA[s] = 2 * (4 * Ib + c4);
    T.offW[s] = ok ? (row * g::LDW + col) : (g::P * g::LDW + 64 * pw + lane);
  }
}

// One D^3 product on the matrix cores, W = Aop^T-layout x X, written row-major to the LDS buffer Wb.
// LDAOP = leading dimension of the A-operand matrix (LDA, or LDX when the stage state itself is the operand).
// Straight-line code: NKP k-pairs, fragments of pair kp+1 are loaded while the MFMAs of pair kp issue.
template <int NB, int LDAOP>
__device__ __forceinline__ void product(const double* __restrict__ Aop, const double* __restrict__ X, double* __restrict__ Wb,
                                        const PTab<NB>& T, int lane) {
  using g = Geo<NB>;
  constexpr int MAXU = g::MAXU, NKP = g::NKP;
  const int r4 = lane >> 4;
  const double* pa = Aop + r4 * LDAOP;
  const double* px = X + r4 * g::LDX;
  d2_t af[2][MAXU], bf[2][2];
  double w[MAXU];
#pragma unroll
  for (int s = 0; s < MAXU; s++) { w[s] = 0.0; af[0][s] = *reinterpret_cast<const d2_t*>(pa + T.colA[s]); }
  bf[0][0] = *reinterpret_cast<const d2_t*>(px + T.colB0);
  bf[0][1] = *reinterpret_cast<const d2_t*>(px + T.colB1);
#pragma unroll
  for (int kp = 0; kp < NKP; kp++) {
    const int cur = kp & 1, nxt = cur ^ 1;
    if (kp + 1 < NKP) {
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
        w[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[cur][s][h], b, w[s], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < MAXU; s++) Wb[T.offW[s]] = w[s];
}

// which LDS buffer a stage's matrix product / mat-vec reads its A operand from
enum : int { OP_R = 0, OP_M = 1, OP_X = 2 };
template <int METHOD, bool FWD>
__host__ __device__ constexpr int stage_op(int j, bool matrix) {
  if (METHOD == VGPA_ODE_RK2) return j == 0 ? ((FWD && matrix) ? OP_X : OP_R) : OP_M;   // Q2: S_k stands in for A_k
  if (METHOD == VGPA_ODE_RK4) return (j == 1 || j == 2) ? OP_M : OP_R;
  return OP_R;                                                                          // Euler, Heun
}
template <int METHOD>
__host__ __device__ constexpr int n_stages() { return METHOD == VGPA_ODE_EULER ? 1 : (METHOD == VGPA_ODE_RK4 ? 4 : 2); }

template <int METHOD, bool FWD, int NB>
__device__ __forceinline__ void p_product_stage(int j, const Lds<NB>& L, const PTab<NB>& T, int lane) {
  using g = Geo<NB>;
  const int op = stage_op<METHOD, FWD>(j, true);
  if (op == OP_X) product<NB, g::LDX>(L.X, L.X, L.W, T, lane);
  else product<NB, g::LDA>(op == OP_M ? L.M : L.R, L.X, L.W, T, lane);
}

// The P role: products for problem A and problem B in alternating phases, one workgroup barrier per phase.
template <int METHOD, bool FWD, int NB, int NPROB>
__device__ __forceinline__ void p_role(int n_steps, bool has_b, const Lds<NB>& LA, const Lds<NB>& LB, int pw, int lane) {
  constexpr int NS = n_stages<METHOD>();
  PTab<NB> T;
  build_ptab<NB>(pw, lane, T);
  __syncthreads();                       // LDS zero-filled
  __syncthreads();                       // E prologue (X, R, xv, Sigma) published
  for (int k = 0; k < n_steps; k++) {
#pragma unroll
    for (int j = 0; j < NS; j++) {
      p_product_stage<METHOD, FWD, NB>(j, LA, T, lane);
      __syncthreads();
      if (NPROB == 2) {
        if (has_b) p_product_stage<METHOD, FWD, NB>(j, LB, T, lane);
        __syncthreads();
      } else {
        __syncthreads();                 // the E phase of the single problem
      }
    }
  }
  if (NPROB == 2) __syncthreads();       // problem B's last element-wise phase
}

// ---- E waves -----------------------------------------------------------------------------------------------------
// Values loaded from HBM before the time loop and only read inside it: make the compiler wait for them HERE.  Otherwise
// its wait-count pass, which cannot see across the loop back-edge that they arrived long ago, puts an s_waitcnt vmcnt(0) in
// front of their first use inside the loop -- behind the prefetches the step has just issued.
__device__ __forceinline__ void settle(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void settle(d2_t& v) { asm volatile("" : "+v"(v)); }

// A row-pair item (p, c) = elements (2p, c) and (2p+1, c) of the D x D state; items are dealt with c fastest over the
// E threads, so a 16-lane group publishes 16 consecutive 16-byte units with one conflict-free ds_write_b128 and its W
// reads are 16 consecutive doubles of one row.
template <int NB>
struct ETab {
  static constexpr int NIT = Geo<NB>::NIT;
  int offX[NIT];    // p * LDX + 2 c
  int offW[NIT];    // (2p) * LDW + c           (row 2p+1: + LDW)
  int offWt[NIT];   // c * LDW + 2p             (16-byte unit W[c][2p], W[c][2p+1])
  int gofs[NIT];    // (2p) * D + c             (row 2p+1: + D)
  int lo[NIT];      // staging: LDS offset of the 16-byte operand unit inside R / M
  int ga0[NIT], ga1[NIT];   // staging: global offsets of its two entries
  unsigned mask;    // bit q: item q exists; bit 8+q: its second row exists; bit 16+q: staging item; bit 24+q: its second entry
};

template <int NB, bool FWD>
__device__ __forceinline__ void build_etab(int D, int te, ETab<NB>& T) {
  using g = Geo<NB>;
  static_assert(g::NIT <= 8, "mask layout");
  const int npair = (D + 1) / 2;
  T.mask = 0u;
#pragma unroll
  for (int q = 0; q < g::NIT; q++) {
    const int e = te + q * kNE;
    const bool ok = e < npair * D;
    const int p = ok ? e / D : 0, c = ok ? e - p * D : 0;
    T.offX[q] = p * g::LDX + 2 * c;
    T.offW[q] = 2 * p * g::LDW + c;
    T.offWt[q] = c * g::LDW + 2 * p;
    T.gofs[q] = 2 * p * D + c;
    if (ok) T.mask |= 1u << q;
    if (ok && 2 * p + 1 < D) T.mask |= 1u << (8 + q);
    // staging item (sp, so) = rows 2sp, 2sp+1 of the operand at column so: forward (operand = A^T) the entries
    // A[so][2sp], A[so][2sp+1] with the pair fastest over the lanes (whole rows of A per npair lanes), backward
    // (operand = A) A[2sp][so], A[2sp+1][so] with the column fastest -- the same decomposition as the state items.
    const int sp = FWD ? (ok ? e % npair : 0) : p, so = FWD ? (ok ? e / npair : 0) : c;
    T.lo[q] = sp * g::LDA + 2 * so;
    T.ga0[q] = FWD ? so * D + 2 * sp : 2 * sp * D + so;
    const bool two = ok && (2 * sp + 1 < D);
    T.ga1[q] = two ? (FWD ? T.ga0[q] + 1 : T.ga0[q] + D) : T.ga0[q];
    if (ok) T.mask |= 1u << (16 + q);
    if (two) T.mask |= 1u << (24 + q);
  }
}

template <int NB>
__device__ __forceinline__ void load_a(const double* __restrict__ A, const ETab<NB>& T, d2_t (&a)[Geo<NB>::NIT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB>::NIT; q++) {
    // (offsets of lanes without an item are 0: always valid; their value is never stored.  The second entry of the last
    //  pair when D is odd is a zero: that operand row only meets zeros.)
    a[q][0] = A[T.ga0[q]];
    const double second = A[T.ga1[q]];
    a[q][1] = ((T.mask >> (24 + q)) & 1u) ? second : 0.0;
  }
}

// After the product of a step's first stage: M <- mid-point of (R, next), R <- next.  R still holds what this thread
// stored a step ago, so the previous operand comes back from LDS instead of living in registers for a whole step.
template <int METHOD, int NB>
__device__ __forceinline__ void stage_operands(const Lds<NB>& L, const ETab<NB>& T, const d2_t (&an)[Geo<NB>::NIT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB>::NIT; q++) {
    if ((T.mask >> (16 + q)) & 1u) {
      if (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4) {
        const d2_t prev = *reinterpret_cast<const d2_t*>(L.R + T.lo[q]);
        d2_t mid;
        mid[0] = 0.5 * (prev[0] + an[q][0]); mid[1] = 0.5 * (prev[1] + an[q][1]);
        *reinterpret_cast<d2_t*>(L.M + T.lo[q]) = mid;
      }
      *reinterpret_cast<d2_t*>(L.R + T.lo[q]) = an[q];
    }
  }
}

// partial mat-vec sums of this problem's stage vector, on the E threads while the P waves are in the product
template <int NB, bool FWD>
__device__ __forceinline__ void matvec_partials(const double* __restrict__ Aop, const Lds<NB>& L, int te) {
  using g = Geo<NB>;
  if (FWD) {          // sum_k Aop[k][i] v[k]: lane = (i, part of the row pairs)
    const int part = te / g::P, i = te - part * g::P;
    if (part < g::NPARTF) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < g::RPP; r++) {
        const int rp = part * g::RPP + r;
        if (rp < g::RP) {
          const d2_t av = *reinterpret_cast<const d2_t*>(Aop + rp * g::LDA + 2 * i);
          const d2_t xk = *reinterpret_cast<const d2_t*>(L.xv + 2 * rp);
          s = __builtin_fma(av[0], xk[0], s);
          s = __builtin_fma(av[1], xk[1], s);
        }
      }
      L.pv[part * g::P + i] = s;
    }
  } else {            // sum_k Aop[i][k] v[k]: lane = (row pair, part of the columns); one read feeds both rows
    constexpr int HP = g::P / 2;
    const int part = te / HP, ip = te - part * HP;
    if (part < g::NPARTB) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int kk = 0; kk < g::KPP; kk++) {
        const int k = part * g::KPP + kk;
        if (k < g::P) {
          const d2_t av = *reinterpret_cast<const d2_t*>(Aop + ip * g::LDA + 2 * k);
          const double xk = L.xv[k];
          s0 = __builtin_fma(av[0], xk, s0);
          s1 = __builtin_fma(av[1], xk, s1);
        }
      }
      d2_t o; o[0] = s0; o[1] = s1;
      *reinterpret_cast<d2_t*>(L.pv + part * g::P + 2 * ip) = o;
    }
  }
}

template <int NB, bool FWD>
__device__ __forceinline__ double matvec_sum(const Lds<NB>& L, int i) {
  using g = Geo<NB>;
  constexpr int NP = FWD ? cmin(g::NPARTF, cdiv(g::RP, g::RPP)) : cmin(g::NPARTB, cdiv(g::P, g::KPP));   // parts holding a sum
  double s = L.pv[i];
#pragma unroll
  for (int q = 1; q < NP; q++) s += L.pv[q * g::P + i];
  return s;
}

// Everything one E group (256 threads) needs to integrate one problem.  has == false: the group only keeps the barrier
// count (odd batch: the last workgroup of a paired launch carries one problem).
template <int METHOD, bool FWD, int NB, bool DENSEJ>
__device__ __forceinline__ void e_role(const OdeArgs& a, int prob, bool has, bool leads, bool trails, const Lds<NB>& L,
                                       double* __restrict__ SIG, bool writes_sig, int te) {
  using g = Geo<NB>;
  constexpr int NIT = g::NIT, NS = n_stages<METHOD>();
  const int D = a.D, DD = D * D, Np = a.Np;
  const double dt = a.dt, h = 0.5 * a.dt;
  const int n_steps = Np - 1;
  if (!has) {
    const int nbar = 2 + 2 * NS * n_steps + 1;
    for (int i = 0; i < nbar; i++) __syncthreads();
    return;
  }
  ETab<NB> T;
  build_etab<NB, FWD>(D, te, T);
  const bool vl = te < D;                                  // this thread carries entry `te` of the vector recursion
  const double* A = a.A + (size_t)prob * a.strideA;
  auto item = [&](int q) { return ((T.mask >> q) & 1u) != 0u; };
  auto row2 = [&](int q) { return ((T.mask >> (8 + q)) & 1u) != 0u; };
  auto opbuf = [&](int op) -> const double* { return op == OP_M ? L.M : L.R; };

  d2_t xk[NIT], acc1[NIT], acc2[NIT], an[NIT];
  double vk = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;           // vector entry and its RK slopes
#pragma unroll
  for (int q = 0; q < NIT; q++) { xk[q] = d2_t{0.0, 0.0}; acc1[q] = acc2[q] = d2_t{0.0, 0.0}; }

  __syncthreads();                                         // LDS zero-filled
  if (FWD) {
    // ------------------------------------------------------------------------------------------ forward: (m, S)
    const double* bb = a.b + (size_t)prob * a.strideB;
    double* mt = a.m + (size_t)prob * Np * D;
    double* st = a.S + (size_t)prob * Np * DD;
#pragma unroll
    for (int q = 0; q < NIT; q++) {
      if (item(q)) {
        xk[q][0] = a.S0[T.gofs[q]];
        xk[q][1] = row2(q) ? a.S0[T.gofs[q] + D] : 0.0;
        st[T.gofs[q]] = xk[q][0];
        if (row2(q)) st[T.gofs[q] + D] = xk[q][1];
        *reinterpret_cast<d2_t*>(L.X + T.offX[q]) = xk[q];
        if (writes_sig) {
          d2_t sg;
          sg[0] = a.Sigma[T.gofs[q]]; sg[1] = row2(q) ? a.Sigma[T.gofs[q] + D] : 0.0;
          *reinterpret_cast<d2_t*>(SIG + 2 * (te + q * kNE)) = sg;
        }
      }
    }
    if (vl) { vk = a.m0[te]; mt[te] = vk; L.xv[te] = vk; }
    load_a<NB>(A, T, an);
#pragma unroll
    for (int q = 0; q < NIT; q++)
      if ((T.mask >> (16 + q)) & 1u) *reinterpret_cast<d2_t*>(L.R + T.lo[q]) = an[q];
    if (Np > 1) load_a<NB>(A + DD, T, an);
    double b0 = vl ? bb[te] : 0.0;
    double b1 = (vl && Np > 1) ? bb[D + te] : 0.0;
#pragma unroll
    for (int q = 0; q < NIT; q++) settle(xk[q]);
    settle(b0); settle(b1); settle(vk);
    __syncthreads();                                       // prologue published
    if (leads) __syncthreads();

    for (int k = 0; k < n_steps; k++) {
      const double b2 = (vl && k + 2 < Np) ? bb[(size_t)(k + 2) * D + te] : 0.0;     // for the next step
      const double bmid = 0.5 * (b0 + b1);
#pragma unroll
      for (int j = 0; j < NS; j++) {
        matvec_partials<NB, true>(opbuf(stage_op<METHOD, true>(j, false)), L, te);
        __syncthreads();
        // ---- element-wise stage j (the P waves are busy with the other problem)
        const bool last = (j == NS - 1);
        double vs = 0.0;
        if (vl) vs = matvec_sum<NB, true>(L, te);
#pragma unroll
        for (int q = 0; q < NIT; q++) {
          if (item(q)) {
            const double w0 = L.W[T.offW[q]], w1 = L.W[T.offW[q] + g::LDW];
            const d2_t wt = *reinterpret_cast<const d2_t*>(L.W + T.offWt[q]);
            const d2_t sg = *reinterpret_cast<const d2_t*>(SIG + 2 * (te + q * kNE));
            d2_t f, xn;
            f[0] = (-w0 - wt[0]) + sg[0]; f[1] = (-w1 - wt[1]) + sg[1];
            if (METHOD == VGPA_ODE_EULER) {
              xk[q] = xk[q] + f * dt; xn = xk[q];
            } else if (METHOD == VGPA_ODE_HEUN) {
              if (j == 0) { acc1[q] = f; xn = xk[q] + f * dt; }
              else { xk[q] = xk[q] + h * (acc1[q] + f); xn = xk[q]; }
            } else if (METHOD == VGPA_ODE_RK2) {
              if (j == 0) xn = xk[q] + h * f;
              else { xk[q] = xk[q] + dt * f; xn = xk[q]; }
            } else {
              if (j == 0) { acc1[q] = f; xn = xk[q] + h * f; }
              else if (j == 1) { acc2[q] = f; xn = xk[q] + h * f; }
              else if (j == 2) { acc2[q] = acc2[q] + f; xn = xk[q] + dt * f; }
              else { xk[q] = xk[q] + dt * (acc1[q] + 2.0 * acc2[q] + f) / 6.0; xn = xk[q]; }
            }
            *reinterpret_cast<d2_t*>(L.X + T.offX[q]) = xn;
          }
        }
        if (vl) {
          double vn;
          if (METHOD == VGPA_ODE_EULER) { vk = vk + (-vs + b0) * dt; vn = vk; }
          else if (METHOD == VGPA_ODE_HEUN) {
            if (j == 0) { v1 = -vs + b0; vn = vk + v1 * dt; }
            else { vk = vk + h * (v1 + (-vs + b1)); vn = vk; }
          } else if (METHOD == VGPA_ODE_RK2) {
            if (j == 0) { vn = vk + h * (-vs + b0); }
            else { vk = vk + dt * (-vs + bmid); vn = vk; }
          } else {
            if (j == 0) { v1 = -vs + b0; vn = vk + h * v1; }
            else if (j == 1) { v2 = -vs + bmid; vn = vk + h * v2; }
            else if (j == 2) { v3 = -vs + bmid; vn = vk + dt * v3; }
            else { vk = vk + dt * (v1 + 2.0 * (v2 + v3) + (-vs + b1)) / 6.0; vn = vk; }
          }
          L.xv[te] = vn;
        }
        if (j == 0) {        // R is free once the first product of the step is over: stage the operands of the rest
          stage_operands<METHOD, NB>(L, T, an);
          if (k + 2 < Np) load_a<NB>(A + (size_t)(k + 2) * DD, T, an);
        }
        if (last) {          // S_{k+1}, m_{k+1} -> HBM
          double* so = st + (size_t)(k + 1) * DD;
#pragma unroll
          for (int q = 0; q < NIT; q++) {
            if (item(q)) {
              so[T.gofs[q]] = xk[q][0];
              if (row2(q)) so[T.gofs[q] + D] = xk[q][1];
            }
          }
          if (vl) mt[(size_t)(k + 1) * D + te] = vk;
        }
        __syncthreads();
      }
      b0 = b1; b1 = b2;
    }
  } else {
    // ------------------------------------------------------------------------------------------ backward: (lam, Psi)
    const double* gm = a.dEm + (size_t)prob * Np * D;
    const double* gs = a.dEs + (size_t)prob * Np * DD;
    double* lam = a.lam + (size_t)prob * Np * D;
    double* psi = a.psi + (size_t)prob * Np * DD;
    d2_t gC[NIT], gN[NIT];
    const bool sparse = !DENSEJ && a.obs_idx;
#pragma unroll
    for (int q = 0; q < NIT; q++) {
      gC[q] = gN[q] = d2_t{0.0, 0.0};
      if (item(q)) {
        const size_t o1 = (size_t)(Np - 1) * DD + T.gofs[q];
        gC[q][0] = gs[o1]; gC[q][1] = row2(q) ? gs[o1 + D] : 0.0;
        if (Np > 1) { gN[q][0] = gs[o1 - DD]; gN[q][1] = row2(q) ? gs[o1 - DD + D] : 0.0; }
        psi[o1] = 0.0;
        if (row2(q)) psi[o1 + D] = 0.0;
        if (writes_sig && !DENSEJ) {
          d2_t js{0.0, 0.0};
          if (a.js_const) { js[0] = a.js_const[T.gofs[q]]; js[1] = row2(q) ? a.js_const[T.gofs[q] + D] : 0.0; }
          *reinterpret_cast<d2_t*>(SIG + 2 * (te + q * kNE)) = js;
        }
      }
    }
    if (vl) lam[(size_t)(Np - 1) * D + te] = 0.0;
    load_a<NB>(A + (size_t)(Np - 1) * DD, T, an);
#pragma unroll
    for (int q = 0; q < NIT; q++)
      if ((T.mask >> (16 + q)) & 1u) *reinterpret_cast<d2_t*>(L.R + T.lo[q]) = an[q];
    if (Np > 1) load_a<NB>(A + (size_t)(Np - 2) * DD, T, an);
    // per-step vectors are fetched one step ahead: g0 = dEsde_dm[t], g1 = dEsde_dm[t-1]; jump of index t-1
    double g0 = vl ? gm[(size_t)(Np - 1) * D + te] : 0.0;
    double g1 = (vl && Np > 1) ? gm[(size_t)(Np - 2) * D + te] : 0.0;
    int n_obs_cur = (sparse && Np > 1) ? a.obs_idx[Np - 2] : -1;
    double jm = 0.0;
    if (Np > 1) {
      if (DENSEJ) { if (vl) jm = a.jm_dense[((size_t)prob * Np + (Np - 2)) * D + te]; }
      else if (vl && n_obs_cur >= 0) jm = a.jm_sparse[((size_t)prob * a.n_obs + n_obs_cur) * D + te];
    }
#pragma unroll
    for (int q = 0; q < NIT; q++) { settle(gC[q]); settle(gN[q]); }
    settle(g0); settle(g1); settle(jm);
    __syncthreads();                                       // prologue published
    if (leads) __syncthreads();

    for (int t = Np - 1; t > 0; t--) {
      const double g2 = (vl && t >= 2) ? gm[(size_t)(t - 2) * D + te] : 0.0;          // for the next step
      const int n_obs_next = (sparse && t >= 2) ? a.obs_idx[t - 2] : -1;
      double jm_next = 0.0;
      if (t >= 2) {
        if (DENSEJ) { if (vl) jm_next = a.jm_dense[((size_t)prob * Np + (t - 2)) * D + te]; }
        else if (vl && n_obs_next >= 0) jm_next = a.jm_sparse[((size_t)prob * a.n_obs + n_obs_next) * D + te];
      }
      const double gmid = 0.5 * (g1 + g0);
#pragma unroll
      for (int j = 0; j < NS; j++) {
        matvec_partials<NB, false>(opbuf(stage_op<METHOD, false>(j, false)), L, te);
        __syncthreads();
        const bool last = (j == NS - 1);
        double vs = 0.0;
        if (vl) vs = matvec_sum<NB, false>(L, te);
#pragma unroll
        for (int q = 0; q < NIT; q++) {
          if (item(q)) {
            const double w0 = L.W[T.offW[q]], w1 = L.W[T.offW[q] + g::LDW];
            const d2_t wt = *reinterpret_cast<const d2_t*>(L.W + T.offWt[q]);
            d2_t w, xn; w[0] = w0; w[1] = w1;
            d2_t js{0.0, 0.0};
            if (last) {      // matrix jump of index t-1, added after the step (euler.py:139-149)
              if (DENSEJ) {
                const double* jp = a.js_dense + ((size_t)prob * Np + (t - 1)) * DD;
                js[0] = jp[T.gofs[q]]; js[1] = row2(q) ? jp[T.gofs[q] + D] : 0.0;
              } else if (n_obs_cur >= 0) {
                js = *reinterpret_cast<const d2_t*>(SIG + 2 * (te + q * kNE));
              }
            }
            const d2_t gmat = 0.5 * (gN[q] + gC[q]);
            if (METHOD == VGPA_ODE_EULER) {
              xk[q] = xk[q] - ((-gC[q] + wt) + w) * dt + js; xn = xk[q];
            } else if (METHOD == VGPA_ODE_HEUN) {
              if (j == 0) { acc1[q] = (-gC[q] + wt) + w; xn = xk[q] - acc1[q] * dt; }
              else { xk[q] = xk[q] - h * (acc1[q] + ((-gN[q] + wt) + w)) + js; xn = xk[q]; }
            } else if (METHOD == VGPA_ODE_RK2) {
              if (j == 0) xn = xk[q] - h * ((-gC[q] + wt) + w);
              else { xk[q] = xk[q] - dt * ((-gmat + wt) + w) + js; xn = xk[q]; }
            } else {
              if (j == 0) { acc1[q] = (-gC[q] + wt) + w; xn = xk[q] - h * acc1[q]; }
              else if (j == 1) { acc2[q] = (-gmat + wt) + w; xn = xk[q] - h * acc2[q]; }
              else if (j == 2) { const d2_t r = (-gmat + wt) + w; acc2[q] = acc2[q] + r; xn = xk[q] - dt * r; }
              else {
                const d2_t r = (-gN[q] + wt) + w;
                xk[q] = xk[q] - dt * (acc1[q] + 2.0 * acc2[q] + r) / 6.0 + js; xn = xk[q];
              }
            }
            *reinterpret_cast<d2_t*>(L.X + T.offX[q]) = xn;
          }
        }
        if (vl) {
          double vn;
          if (METHOD == VGPA_ODE_EULER) { vk = vk - (-g0 + vs) * dt + jm; vn = vk; }
          else if (METHOD == VGPA_ODE_HEUN) {
            if (j == 0) { v1 = -g0 + vs; vn = vk - v1 * dt; }
            else { vk = vk - h * (v1 + (-g1 + vs)) + jm; vn = vk; }
          } else if (METHOD == VGPA_ODE_RK2) {
            if (j == 0) { vn = vk - h * (-g0 + vs); }
            else { vk = vk - dt * (-gmid + vs) + jm; vn = vk; }
          } else {
            if (j == 0) { v1 = -g0 + vs; vn = vk - h * v1; }
            else if (j == 1) { v2 = -gmid + vs; vn = vk - h * v2; }
            else if (j == 2) { v3 = -gmid + vs; vn = vk - dt * v3; }
            else { vk = vk - dt * (v1 + 2.0 * (v2 + v3) + (-g1 + vs)) / 6.0 + jm; vn = vk; }
          }
          L.xv[te] = vn;
        }
        if (j == 0) {
          stage_operands<METHOD, NB>(L, T, an);
          if (t >= 2) load_a<NB>(A + (size_t)(t - 2) * DD, T, an);
        }
        if (last) {          // Psi_{t-1}, lam_{t-1} -> HBM; rotate G
          double* po = psi + (size_t)(t - 1) * DD;
#pragma unroll
          for (int q = 0; q < NIT; q++) {
            if (item(q)) {
              po[T.gofs[q]] = xk[q][0];
              if (row2(q)) po[T.gofs[q] + D] = xk[q][1];
              gC[q] = gN[q];
              if (t >= 2) {
                const size_t o2 = (size_t)(t - 2) * DD + T.gofs[q];
                gN[q][0] = gs[o2]; gN[q][1] = row2(q) ? gs[o2 + D] : 0.0;
              }
            }
          }
          if (vl) lam[(size_t)(t - 1) * D + te] = vk;
        }
        __syncthreads();
      }
      g0 = g1; g1 = g2; jm = jm_next; n_obs_cur = n_obs_next;
    }
  }
  if (trails) __syncthreads();
}

// =================================================================================================================
// Workgroup = 4 P waves + 4 E waves per problem.  Problems of workgroup w: the first `npair` workgroups take two
// (2w, 2w+1), the others one (npair + w).
template <int METHOD, bool FWD, int NB, int NPROB, bool DENSEJ>
__global__ void __launch_bounds__(64 * (kNPW + 4 * NPROB)) k_ode_pe(OdeArgs a, int npair) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = Geo<NB>;
  constexpr int NT = 64 * (kNPW + 4 * NPROB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = (int)blockIdx.x;
  const bool paired = (NPROB == 2) && wg < npair;
  const int prob_a = (NPROB == 2) ? (paired ? 2 * wg : npair + wg) : wg;
  const int prob_b = prob_a + 1;
  Lds<NB> LA, LB;
  LA.carve(smem);
  LB.carve(smem + (NPROB == 2 ? g::PROB : 0));
  double* SIG = smem + (size_t)NPROB * g::PROB;
  for (int i = tid; i < (int)g::lds_doubles(NPROB); i += NT) smem[i] = 0.0;
  if (wave < kNPW) {
    p_role<METHOD, FWD, NB, NPROB>(a.Np - 1, paired, LA, LB, wave, lane);
  } else if (wave < kNPW + 4) {
    // problem A: its element-wise phase is the one in which the P waves work for B (or idle); it waits out B's last phase
    e_role<METHOD, FWD, NB, DENSEJ>(a, prob_a, true, false, NPROB == 2, LA, SIG, true, tid - 64 * kNPW);
  } else {
    e_role<METHOD, FWD, NB, DENSEJ>(a, prob_b, paired, true, false, LB, SIG, false, tid - 64 * (kNPW + 4));
  }
}

template <int METHOD, bool FWD, int NB, int NPROB>
hipError_t launch_nb_p(const OdeArgs& a, int nwg, int npair, hipStream_t st) {
  constexpr size_t lds = Geo<NB>::lds_doubles(NPROB) * sizeof(double);
  static_assert(lds <= 160 * 1024, "LDS budget");
  const bool dense = !FWD && a.js_dense;
  auto kern = dense ? k_ode_pe<METHOD, FWD, NB, NPROB, true> : k_ode_pe<METHOD, FWD, NB, NPROB, false>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * (kNPW + 4 * NPROB)), lds, st, a, npair);
  return hipGetLastError();
}

int device_cu_count();   // ode_mfma.hip

// One problem per workgroup while every problem can have a CU of its own; beyond that, pairs -- as few as needed to fit
// the batch on the chip in one wave of workgroups, everything paired from two problems per CU on.
template <int METHOD, bool FWD, int NB>
hipError_t launch_nb(const OdeArgs& a, hipStream_t st) {
  const int B = a.batch;
  if constexpr (NB <= kMaxPairNB) {
    const int ncu = device_cu_count();
    const bool pairs = a.pair_mode == 2 ? (B >= 2) : (a.pair_mode == 1 ? false : B > ncu);
    if (pairs) {
      int nwg = (B + 1) / 2;
      if (a.pair_mode != 2 && nwg < ncu) nwg = ncu;
      return launch_nb_p<METHOD, FWD, NB, 2>(a, nwg, B - nwg, st);
    }
  }
  return launch_nb_p<METHOD, FWD, NB, 1>(a, B, 0, st);
}

}  // namespace mfma

// One instantiation set per stepper (defined in ode_mfma_m<METHOD>.hip).
template <int METHOD> bool mfma_method_supported(int nb);
template <int METHOD> hipError_t mfma_method_launch(bool fwd, const OdeArgs& a, hipStream_t st);

}  // namespace vgpa
