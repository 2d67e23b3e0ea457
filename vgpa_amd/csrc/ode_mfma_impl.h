// Symmetric fast path of the time-stepping kernels: fp64 matrix cores, role-specialised waves, one workgroup per problem.
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
//   * One workgroup (4 P + 4 E waves, one of each per SIMD) per problem; the phases alternate product / element-wise
//     stage with ONE workgroup barrier each.  Also built and measured in round 2, not kept (EXPERIMENTS.md s.1): two problems
//     per workgroup in alternating phases -- with separate E waves per problem and with every E wave serving both --
//     (the matrix pipe has work in every phase, but LDS traffic and the fp64 VALU work of the other problem's element-wise
//     stage stretch the product by what the overlap saves: +4 % forward, slower backward), and counters in LDS instead of
//     barriers (products back to back, but polling and the longer hand-off latency cost more than the drained pipeline).
// LDS per problem (D = 40): stage state X 15 KB + two A-operand buffers (start/end point R, mid-point M) 26 KB + W 13 KB
// + vectors + the constant matrix jump 13 KB: 82 KB.
#pragma once
#include "vgpa_internal.h"
#include <cstdlib>

namespace vgpa {
namespace sym {
// VGPA_SYM_RUNS=1 keeps the run layout of the symmetric-unit kernels where the fragment cover exists (comparison runs)
inline bool runs_only_env() {
  static const bool r = [] { const char* e = getenv("VGPA_SYM_RUNS"); return e && e[0] == '1'; }();
  return r;
}
// the backward kernels that can store Q''_t = Sigma^-1 A_t - 2 Psi_t instead of Psi_t (OdeArgs::q_on): the fragment-cover
// kernels (ode_sym_impl.h, 33 <= D <= 40) of the mid-point methods with sparse jumps
inline bool stores_q(int method, int D) {
  return (method == VGPA_ODE_RK2 || method == VGPA_ODE_RK4) && D >= 33 && D <= 40 && !runs_only_env();
}
// the backward kernel that assembles the gradient on its helper waves (OdeArgs::grad_on; k_ode_sym, GF): the fragment-cover kernels
// of RK4 (VGPA_SYM_COVER=op, the outer-product experiment, has no helper waves)
inline bool fuses_grad(int method, int D) {
#ifdef VGPA_EXPERIMENTS
  static const bool op = [] { const char* e = getenv("VGPA_SYM_COVER"); return e && e[0] == 'o' && e[1] == 'p' && !e[2]; }();
#else
  const bool op = false;
#endif
  static const bool off = [] { const char* e = getenv("VGPA_FUSED_GRAD"); return e && e[0] == '0'; }();
  return method == VGPA_ODE_RK4 && stores_q(method, D) && !op && !off;
}
}  // namespace sym
}  // namespace vgpa

namespace vgpa {
namespace mfma {

constexpr int kMaxNB = 11;       // D <= 44
constexpr int kNPW = 4;          // P waves (one per SIMD)
constexpr int kNE = 256;         // E threads (4 waves, one per SIMD)

typedef double d2_t __attribute__((ext_vector_type(2)));

// Per-lane HBM accesses go through a wave-uniform base pointer + an UNSIGNED 32-bit byte offset: the instruction then
// takes the base from SGPRs and the offset from one VGPR.  With 64-bit per-lane indices the compiler precomputes every
// (item, row, array) address as a VGPR pair outside the time loop -- 30-60 registers that decide whether the kernel spills.
__device__ __forceinline__ double ldg(const double* base, unsigned off8) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off8);
}
using vgpa::ldu;
__device__ __forceinline__ void stg(double* base, unsigned off8, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off8) = v;
}

// Diagnostic build only (tools/ubench/ode_pe_stamp.hip): per-segment cycle sums of one P wave and one E wave of workgroup 0.
// Never defined in the product build, so no stamp executes there.
#ifdef VGPA_STAMPS_ROLE
__device__ long long g_stamp_role[4][16];
#endif
#ifdef VGPA_STAMPS
__device__ long long g_stamp[4][8];
#define VGPA_STAMP(role, i)                                                                   \
  do {                                                                                        \
    const long long t_ = clock64();                                                           \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_stamp[role][i] += t_ - stamp_prev_;     \
    stamp_prev_ = t_;                                                                         \
  } while (0)
#define VGPA_STAMP_DECL long long stamp_prev_ = clock64()
#else
#define VGPA_STAMP(role, i) do {} while (0)
#define VGPA_STAMP_DECL do {} while (0)
#endif

// ---- dealing MFMA units to (P wave, slot) -----------------------------------------------------------------------
// A "unit" is one MFMA accumulator = four 4x4 output blocks: (I, J = 4q..4q+3) for the full 16-column groups, and the
// left-over column blocks of several block-rows packed together (D = 40: 100 blocks = 25 units = 250 MFMAs per product).
// Units are numbered group-major: NQ full column groups of NB units each, then NLEFT left-over units.  All units of a
// group share one B fragment.  Every wave loads TWO B fragments per k-step and slot s uses the first one when s < S1 and
// the second one otherwise -- a compile-time choice, the same for all waves; every wave issues all MAXU slots (slots
// without a unit multiply garbage into a trash word), so the product takes MAXU x KKE MFMAs on every SIMD and the deal
// only has to minimise MAXU.  Per wave: the rest of the current group if it fits one of the two slot ranges (the smaller
// one that takes it), the other range from the next group; else both ranges from the current group.
struct WaveDeal { int uA, nA, gA, uB, nB, gB; };   // first unit / count / group of the A-range and of the B-range

__host__ __device__ constexpr int deal_group_size(int nb, int nq, int nleft, int g) { return g < nq ? nb : (g == nq ? nleft : 0); }

__host__ __device__ constexpr WaveDeal deal_units(int nb, int nq, int nleft, int maxu, int s1, int nw, int want, bool* done) {
  const int ngroups = nq + (nleft ? 1 : 0);
  const int s2 = maxu - s1;
  int g = 0, off = 0;
  WaveDeal res{0, 0, 0, 0, 0, 0};
  for (int w = 0; w < nw; w++) {
    while (g < ngroups && off >= deal_group_size(nb, nq, nleft, g)) { g++; off = 0; }
    WaveDeal d{0, 0, 0, 0, 0, 0};
    if (g < ngroups) {
      const int rem = deal_group_size(nb, nq, nleft, g) - off;
      if (rem > s1 && rem > s2) {               // both ranges from this group
        const int na = rem < s1 ? rem : s1;
        const int nbb = (rem - na) < s2 ? (rem - na) : s2;
        d = WaveDeal{g * nb + off, na, g, g * nb + off + na, nbb, g};
        off += na + nbb;
      } else {                                  // the rest of the group in one range, the next group in the other
        const bool in_a = (rem <= s1) && (s1 <= s2 || rem > s2);
        const int u_first = g * nb + off, g_first = g;
        off += rem;
        while (g < ngroups && off >= deal_group_size(nb, nq, nleft, g)) { g++; off = 0; }
        int n2 = 0, u2 = 0, g2 = g_first;
        if (g < ngroups) {
          const int cap = in_a ? s2 : s1;
          const int rem2 = deal_group_size(nb, nq, nleft, g) - off;
          n2 = rem2 < cap ? rem2 : cap;
          u2 = g * nb + off; g2 = g;
          off += n2;
        }
        d = in_a ? WaveDeal{u_first, rem, g_first, u2, n2, g2} : WaveDeal{u2, n2, g2, u_first, rem, g_first};
      }
    }
    if (w == want) res = d;
  }
  while (g < ngroups && off >= deal_group_size(nb, nq, nleft, g)) { g++; off = 0; }
  if (done) *done = (g >= ngroups);
  return res;
}

__host__ __device__ constexpr bool deal_fits(int nb, int nq, int nleft, int maxu, int s1, int nw) {
  bool ok = false;
  (void)deal_units(nb, nq, nleft, maxu, s1, nw, 0, &ok);
  return ok;
}

// smallest MAXU (and, for it, the most even split S1) for which the deal covers every unit; want_s1 selects the result
__host__ __device__ constexpr int deal_pick(int nb, int nq, int nleft, int nu, int nw, bool want_s1) {
  for (int m = (nu + nw - 1) / nw; m <= nu; m++)
    for (int s1 = m / 2; s1 >= 0; s1--)
      if (deal_fits(nb, nq, nleft, m, s1, nw)) return want_s1 ? s1 : m;
  return want_s1 ? 0 : nu;
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
  static constexpr int MAXU = deal_pick(NB, NQ, NLEFT, NU, kNPW, false);   // unit slots per P wave
  static constexpr int S1 = deal_pick(NB, NQ, NLEFT, NU, kNPW, true);      // slots [0,S1): B fragment 0, [S1,MAXU): fragment 1
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
  static constexpr int NIT = cdiv((P / 2) * P, 64 * kNPW);   // 16-byte operand units per P lane (staging)
  // per-problem LDS regions (doubles)
  static constexpr int XS = RP * LDX;
  static constexpr int AS = RP * LDA;
  static constexpr int WS = P * LDW + 64 * kNPW + 2 * kNE;   // + one trash double per P lane + one 16-byte unit per E thread
  static constexpr int XV = 4 * KKE + 4;                  // stage vector, padded like the operand rows
  static constexpr int PV = cmax(cmin(kNE / P, RP), cmin(kNE / (P / 2), P)) * P;   // partial mat-vec sums, sized for 256 E threads
  static constexpr int PROB = XS + 2 * AS + WS + XV + PV;
  static constexpr int SIGS = P * P;                      // Sigma / constant matrix jump in item layout
  static constexpr size_t LDS_DOUBLES = (size_t)PROB + SIGS + 2 * kNE;
};

// what depends on the number NE of E threads of a problem
template <int NB, int NE>
struct EGeo {
  using g = Geo<NB>;
  static constexpr int NIT = cdiv((g::P / 2) * g::P, NE);   // row-pair items per E thread
  // mat-vec partial sums on the E threads: forward lane = (column i, part of the row pairs), backward lane = (row pair,
  // part of the columns)
  static constexpr int NPARTF = cmin(NE / g::P, g::RP);
  static constexpr int RPP = cdiv(g::RP, NPARTF);
  static constexpr int NPARTB = cmin(NE / (g::P / 2), g::P);
  static constexpr int KPP = cdiv(g::P, NPARTB);
  static_assert(NPARTF >= 1 && NPARTB >= 1, "too few E threads for this D");
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

// Values loaded from HBM before the time loop and only read inside it: make the compiler wait for them HERE.  Otherwise
// its wait-count pass, which cannot see across the loop back-edge that they arrived long ago, puts an s_waitcnt vmcnt(0) in
// front of their first use inside the loop -- behind the prefetches the step has just issued.
__device__ __forceinline__ void settle(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void settle(d2_t& v) { asm volatile("" : "+v"(v)); }

// ---- P waves -----------------------------------------------------------------------------------------------------
template <int NB>
struct PTab {
  static constexpr int MAXU = Geo<NB>::MAXU;
  int colA[MAXU];   // 2*(4*I_b + (l&3))
  int colB0, colB1; // B-fragment columns of the group feeding slots [0,S1) / slots [S1,MAXU)
  int offW[MAXU];   // row * LDW + col of the element this lane's accumulator holds (trash for lanes without one)
  unsigned gofs[MAXU];   // 8 (row * D + col): BYTE offset of that element inside a D x D matrix in HBM (0 without one)
  unsigned own;     // bit s: slot s holds a real matrix element (row, col < D)
};

template <int NB>
__device__ __forceinline__ void build_ptab(int D, int pw, int lane, PTab<NB>& T) {
  using g = Geo<NB>;
  const int b = (lane >> 2) & 3, r4 = lane >> 4, c4 = lane & 3;
  const WaveDeal deal = deal_units(g::NB, g::NQ, g::NLEFT, g::MAXU, g::S1, kNPW, pw, nullptr);
  constexpr int rem = g::REM ? g::REM : 1;
  auto group_col = [&](int grp) { return (grp < g::NQ) ? (16 * grp + (lane & 15)) : (4 * (4 * g::NQ + b % rem) + c4); };
  T.colB0 = 2 * group_col(deal.gA);
  T.colB1 = 2 * group_col(deal.gB);
  T.own = 0u;
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
    const bool own = ok && row < D && col < D;
    T.gofs[s] = own ? 8u * (unsigned)(row * D + col) : 0u;
    if (own) T.own |= 1u << s;
  }
}

// One D^3 product on the matrix cores, W = Aop^T-layout x X, written row-major to the LDS buffer Wb.
// LDAOP = leading dimension of the A-operand matrix (LDA, or LDX when the stage state itself is the operand).
// Straight-line code: NKP k-pairs, fragments of pair kp+1 are loaded while the MFMAs of pair kp issue.  (A hand-pinned
// variant -- three buffers, all reads of pair kp+2 in one block in front of the MFMAs of pair kp -- was 10 % slower: nine
// back-to-back ds_read_b128 keep the wave from issuing MFMAs for ~70 cycles; the compiler's interleaving is kept.)
// The accumulators start from w0 = minus half the stage's symmetric forcing term (Sigma forward, dEsde_dS backward), so
// that what the E waves read is V = W -+ F/2 and the stage slope is just -(V + V^T) / +(V + V^T): the forcing term never
// enters the element-wise waves (no LDS read, no registers there).
template <int NB, int LDAOP>
__device__ __forceinline__ void product(const double* __restrict__ Aop, const double* __restrict__ X, double* __restrict__ Wb,
                                        const PTab<NB>& T, int lane, const double (&w0)[Geo<NB>::MAXU]) {
  using g = Geo<NB>;
  constexpr int MAXU = g::MAXU, NKP = g::NKP;
  const int r4 = lane >> 4;
  const double* pa = Aop + r4 * LDAOP;
  const double* px = X + r4 * g::LDX;
  d2_t af[2][MAXU], bf[2][2];
  double w[MAXU];
#pragma unroll
  for (int s = 0; s < MAXU; s++) { w[s] = w0[s]; af[0][s] = *reinterpret_cast<const d2_t*>(pa + T.colA[s]); }
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

// Which LDS buffer a stage's matrix product / mat-vec reads its A operand from.  Two buffers per problem: R holds the
// operand of the step's first stage.  While the P waves are in THAT product the E waves fill M -- the mid-point
// 0.5 (A_k + A_{k+1}) (RK2, RK4) or A_{k+1} itself (Heun) -- and one product later, when nothing reads R any more, they
// overwrite R with A_{k+1} for the step's last stage and the next step's first one.  Euler has one stage per step: its
// operand alternates between R (even steps) and M (odd steps).  All of that happens beside a product, never between two.
enum : int { OP_R = 0, OP_M = 1, OP_X = 2 };
template <int METHOD, bool FWD>
__host__ __device__ constexpr int stage_op(int j, bool matrix, int step) {
  if (METHOD == VGPA_ODE_EULER) return (step & 1) ? OP_M : OP_R;
  if (METHOD == VGPA_ODE_HEUN) return j == 0 ? OP_R : OP_M;
  if (METHOD == VGPA_ODE_RK2) return j == 0 ? ((FWD && matrix) ? OP_X : OP_R) : OP_M;   // Q2: S_k stands in for A_k
  return (j == 1 || j == 2) ? OP_M : OP_R;                                              // RK4
}
template <int METHOD>
__host__ __device__ constexpr int n_stages() { return METHOD == VGPA_ODE_EULER ? 1 : (METHOD == VGPA_ODE_RK4 ? 4 : 2); }

// forcing term of one problem in the P lanes: forward the constant Sigma (fc; fn unused), backward G_t (fc) and G_{t-1}
// (fn) of the current step, element per accumulator slot
template <int METHOD, bool FWD, int NB>
__device__ __forceinline__ void stage_w0(int j, const PTab<NB>& T, const double (&fc)[Geo<NB>::MAXU],
                                         const double (&fn)[Geo<NB>::MAXU], double (&w0)[Geo<NB>::MAXU]) {
#pragma unroll
  for (int s = 0; s < Geo<NB>::MAXU; s++) {
    double f;
    if (FWD) f = fc[s];
    else if (METHOD == VGPA_ODE_EULER) f = fc[s];
    else if (METHOD == VGPA_ODE_HEUN) f = j == 0 ? fc[s] : fn[s];
    else if (METHOD == VGPA_ODE_RK2) f = j == 0 ? fc[s] : 0.5 * (fn[s] + fc[s]);
    else f = j == 0 ? fc[s] : (j == 3 ? fn[s] : 0.5 * (fn[s] + fc[s]));
    w0[s] = ((T.own >> s) & 1u) ? -0.5 * f : 0.0;
  }
}

template <int METHOD, bool FWD, int NB>
__device__ __forceinline__ void p_product_stage(int j, int step, const Lds<NB>& L, const PTab<NB>& T, int lane,
                                                const double (&fc)[Geo<NB>::MAXU], const double (&fn)[Geo<NB>::MAXU]) {
  using g = Geo<NB>;
  double w0[g::MAXU];
  stage_w0<METHOD, FWD, NB>(j, T, fc, fn, w0);
  const int op = stage_op<METHOD, FWD>(j, true, step);
  if (op == OP_X) product<NB, g::LDX>(L.X, L.X, L.W, T, lane, w0);
  else product<NB, g::LDA>(op == OP_M ? L.M : L.R, L.X, L.W, T, lane, w0);
}

// ---- A(t): HBM -> registers -> LDS operand buffers, by the P waves -------------------------------------------------------
// The k-pair interleaved operand layout keeps rows 2p and 2p+1 of the operand in one 16-byte unit per column.  A staging
// item is such a unit: (sp, so) = rows 2sp, 2sp+1 of the operand at column so -- forward (operand = A^T) the entries
// A[so][2sp], A[so][2sp+1] with the pair fastest over the 256 P lanes (whole rows of A per npair lanes: contiguous HBM
// reads), backward (operand = A) A[2sp][so], A[2sp+1][so] with the column fastest.  Either way a 16-lane group stores 16
// units with one conflict-free ds_write_b128.  The P waves own no Runge-Kutta state, so they have the registers to keep
// A of the next grid point in flight for a whole step (per problem), and their matrix-core stream has the issue slots.
template <int NB>
struct STab {
  static constexpr int NIT = Geo<NB>::NIT;
  int lo[NIT];              // LDS offset of the unit inside R / M (lanes without an item: their trash unit behind W)
  unsigned ga0[NIT], ga1[NIT];   // global BYTE offsets of its two entries
};

template <int NB, bool FWD>
__device__ __forceinline__ void build_stab(int D, int tp, STab<NB>& T) {
  using g = Geo<NB>;
  const int npair = (D + 1) / 2;
  // (lanes without an item use the trash unit of the E thread of the same index: garbage either way)
#pragma unroll
  for (int q = 0; q < g::NIT; q++) {
    const int e = tp + q * 64 * kNPW;
    const bool ok = e < npair * D;
    const int sp = ok ? (FWD ? e % npair : e / D) : 0, so = ok ? (FWD ? e / npair : e - (e / D) * D) : 0;
    T.lo[q] = ok ? sp * g::LDA + 2 * so : -1;
    T.ga0[q] = 8u * (unsigned)(FWD ? so * D + 2 * sp : 2 * sp * D + so);
    const bool two = ok && (2 * sp + 1 < D);
    T.ga1[q] = two ? (FWD ? T.ga0[q] + 8u : T.ga0[q] + 8u * (unsigned)D) : T.ga0[q];
  }
}

// A(t) for one step, HBM -> registers.  Nothing may depend on the loaded values here: they are consumed a step later, and a
// select behind the load would put an s_waitcnt vmcnt in front of it -- the full HBM latency, inside a phase.  Lanes
// without an item load element 0 (always valid; never stored).  When D is odd the second entry of the last pair repeats
// the first one: it lands in operand row D, which only ever meets the zero rows / entries beyond D of the stage state.
template <int NB>
__device__ __forceinline__ void load_a(const double* __restrict__ A, const STab<NB>& T, d2_t (&a)[Geo<NB>::NIT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB>::NIT; q++) {
    a[q][0] = ldg(A, T.ga0[q]);
    a[q][1] = ldg(A, T.ga1[q]);
  }
}

template <int NB>
__device__ __forceinline__ void store_a(double* __restrict__ buf, double* __restrict__ trash, const STab<NB>& T,
                                        const d2_t (&v)[Geo<NB>::NIT]) {
#pragma unroll
  for (int q = 0; q < Geo<NB>::NIT; q++) *reinterpret_cast<d2_t*>(T.lo[q] >= 0 ? buf + T.lo[q] : trash) = v[q];
}

// What follows the product of stage j of step `step`, done while the E waves are in their element-wise phase: see
// stage_op.  `an` = A at the step's end point.
template <int METHOD, bool FWD, int NB>
__device__ __forceinline__ void p_stage_after(int j, int step, int n_steps, const double* __restrict__ A, int DD, int Np,
                                              const Lds<NB>& L, const STab<NB>& T, d2_t (&an)[Geo<NB>::NIT], int tp) {
  using g = Geo<NB>;
  constexpr int NS = n_stages<METHOD>();
  constexpr int JSEC = NS > 1 ? 1 : 0;
  double* trash = L.W + g::P * g::LDW + 64 * kNPW + 2 * tp;
  if (j == 0) {
    double* dst = (METHOD == VGPA_ODE_EULER && (step & 1)) ? L.R : L.M;     // Euler: the buffer the next step reads
    if (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4) {
      // R still holds what this lane stored a step ago: the start-point operand comes back from LDS
      d2_t mid[g::NIT];
#pragma unroll
      for (int q = 0; q < g::NIT; q++) mid[q] = *reinterpret_cast<const d2_t*>(T.lo[q] >= 0 ? L.R + T.lo[q] : trash);
#pragma unroll
      for (int q = 0; q < g::NIT; q++) { mid[q][0] = 0.5 * (mid[q][0] + an[q][0]); mid[q][1] = 0.5 * (mid[q][1] + an[q][1]); }
      store_a<NB>(dst, trash, T, mid);
    } else {
      store_a<NB>(dst, trash, T, an);
    }
  }
  if (j == JSEC) {
    if (NS > 1) store_a<NB>(L.R, trash, T, an);
    if (step + 2 <= n_steps) load_a<NB>(A + (size_t)(FWD ? step + 2 : Np - 1 - (step + 2)) * DD, T, an);
  }
}

// The P role: one product per stage; the operand staging is done while the E waves are in their element-wise phase.
template <int METHOD, bool FWD, int NB>
__device__ __forceinline__ void p_role(const OdeArgs& a, int prob, const Lds<NB>& L, int pw, int lane) {
  constexpr int NS = n_stages<METHOD>();
  using g = Geo<NB>;
  const int Np = a.Np, DD = a.D * a.D, n_steps = a.Np - 1;
  const int tp = 64 * pw + lane;
  PTab<NB> T;
  build_ptab<NB>(a.D, pw, lane, T);
  STab<NB> S;
  build_stab<NB, FWD>(a.D, tp, S);
  const double* A = a.A + (size_t)prob * a.strideA;
  d2_t an[g::NIT];
  // forcing term per accumulator slot: forward Sigma, backward G_t / G_{t-1}, requested from HBM one step ahead
  const double* G = FWD ? a.Sigma : a.dEs + (size_t)prob * Np * DD;
  double fc[g::MAXU], fn[g::MAXU];
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) {
    const size_t last = FWD ? 0 : (size_t)(Np - 1) * DD, prev = FWD ? 0 : (size_t)(Np > 1 ? Np - 2 : 0) * DD;
    fc[s] = ldg(G + last, T.gofs[s]);
    fn[s] = FWD ? 0.0 : ldg(G + prev, T.gofs[s]);
  }
  __syncthreads();                       // LDS zero-filled
  {  // operands of the first step: R <- A at the first grid point of the sweep; A at the second one stays in flight
    const size_t t0 = FWD ? 0 : (size_t)(Np - 1), t1 = FWD ? 1 : (size_t)(Np - 2);
    load_a<NB>(A + t0 * DD, S, an);
    store_a<NB>(L.R, L.W + g::P * g::LDW + 64 * kNPW + 2 * tp, S, an);
    if (n_steps >= 1) load_a<NB>(A + t1 * DD, S, an);
  }
#pragma unroll
  for (int s = 0; s < g::MAXU; s++) { settle(fc[s]); settle(fn[s]); }
  __syncthreads();                       // prologue (X, R, xv, constant jump) published
  VGPA_STAMP_DECL;
  for (int k = 0; k < n_steps; k++) {
#pragma unroll
    for (int j = 0; j < NS; j++) {
      p_product_stage<METHOD, FWD, NB>(j, k, L, T, lane, fc, fn);
      if (!FWD && j == NS - 1) {         // G of the next step: G_{t-1} becomes the start point, G_{t-2} is requested
        const size_t nxt = (size_t)(Np - 1 - (k + 2) >= 0 ? Np - 1 - (k + 2) : 0) * DD;
#pragma unroll
        for (int s = 0; s < g::MAXU; s++) { fc[s] = fn[s]; fn[s] = ldg(G + nxt, T.gofs[s]); }
      }
      VGPA_STAMP(0, 0);                  // product
      __syncthreads();
      VGPA_STAMP(0, 1);                  // barrier
      p_stage_after<METHOD, FWD, NB>(j, k, n_steps, A, DD, Np, L, S, an, tp);
      VGPA_STAMP(0, 2);                  // operand staging
      __syncthreads();
      VGPA_STAMP(0, 3);                  // barrier (the element-wise phase)
    }
  }
}

// ---- E waves -----------------------------------------------------------------------------------------------------
// A row-pair item (p, c) = elements (2p, c) and (2p+1, c) of the D x D state; items are dealt with c fastest over the
// E threads, so a 16-lane group publishes 16 consecutive 16-byte units with one conflict-free ds_write_b128 and its W
// reads are 16 consecutive doubles of one row.
template <int NB, int NE>
struct ETab {
  static constexpr int NIT = EGeo<NB, NE>::NIT;
  int offX[NIT];    // p * LDX + 2 c
  int offW[NIT];    // (2p) * LDW + c           (row 2p+1: + LDW)
  int offWt[NIT];   // c * LDW + 2p             (16-byte unit W[c][2p], W[c][2p+1])
  unsigned gofs[NIT];   // 8 ((2p) * D + c): BYTE offset inside a D x D matrix (row 2p+1: + 8 D)
  unsigned mask;    // bit q: item q exists; bit 8+q: its second row exists
};

template <int NB, int NE>
__device__ __forceinline__ void build_etab(int D, int te, ETab<NB, NE>& T) {
  using g = Geo<NB>;
  constexpr int NIT = EGeo<NB, NE>::NIT;
  static_assert(NIT <= 8, "mask layout");
  const int npair = (D + 1) / 2;
  // LDS offsets of threads without an item point at the thread's private 16-byte trash unit behind W (offsets relative
  // to X for offX, to W for offW / offWt, to R / M for lo), so the element-wise stage and the staging need no branches.
  const int trashW = g::P * g::LDW + 64 * kNPW + 2 * te;
  const int trashX = g::XS + 2 * g::AS + trashW;           // X + trashX == W + trashW
  T.mask = 0u;
#pragma unroll
  for (int q = 0; q < NIT; q++) {
    const int e = te + q * NE;
    const bool ok = e < npair * D;
    const int p = ok ? e / D : 0, c = ok ? e - p * D : 0;
    T.offX[q] = ok ? p * g::LDX + 2 * c : trashX;
    T.offW[q] = 2 * p * g::LDW + c;
    T.offWt[q] = c * g::LDW + 2 * p;
    T.gofs[q] = 8u * (unsigned)(2 * p * D + c);
    if (ok) T.mask |= 1u << q;
    if (ok && 2 * p + 1 < D) T.mask |= 1u << (8 + q);
  }
}

// partial mat-vec sums of this problem's stage vector, on the E threads while the P waves are in the product.
// Branch-free and with affine addresses (one base register + immediate offsets): threads beyond the last part redo part
// 0 and drop the result into their trash unit; a part that runs past the last row pair / column reads whatever follows in
// LDS (the next operand rows: finite numbers) against the zero entries of the stage vector beyond D.
template <int NB, int NE, bool FWD>
__device__ __forceinline__ void matvec_partials(const double* __restrict__ Aop, const Lds<NB>& L, int te) {
  using g = Geo<NB>;
  using eg = EGeo<NB, NE>;
  double* trash = L.W + g::P * g::LDW + 64 * kNPW + 2 * te;
  if (FWD) {          // sum_k Aop[k][i] v[k]: lane = (i, part of the row pairs)
    constexpr int NP = cmin(eg::NPARTF, cdiv(g::RP, eg::RPP));
    static_assert(2 * NP * eg::RPP <= g::XV, "stage vector padding");
    const int part0 = te / g::P, i = te - part0 * g::P;
    const bool act = part0 < NP;
    const int part = act ? part0 : 0;
    const double* pa = Aop + part * eg::RPP * g::LDA + 2 * i;
    const double* pv = L.xv + 2 * part * eg::RPP;
    d2_t av[eg::RPP], xk[eg::RPP];
#pragma unroll
    for (int r = 0; r < eg::RPP; r++) {
      av[r] = *reinterpret_cast<const d2_t*>(pa + r * g::LDA);
      xk[r] = *reinterpret_cast<const d2_t*>(pv + 2 * r);
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < eg::RPP; r++) {
      s = __builtin_fma(av[r][0], xk[r][0], s);
      s = __builtin_fma(av[r][1], xk[r][1], s);
    }
    *(act ? L.pv + part * g::P + i : trash) = s;
  } else {            // sum_k Aop[i][k] v[k]: lane = (row pair, part of the columns); one read feeds both rows
    constexpr int HP = g::P / 2;
    constexpr int NP = cmin(eg::NPARTB, cdiv(g::P, eg::KPP));
    static_assert(NP * eg::KPP <= g::XV, "stage vector padding");
    const int part0 = te / HP, ip = te - part0 * HP;
    const bool act = part0 < NP;
    const int part = act ? part0 : 0;
    const double* pa = Aop + ip * g::LDA + 2 * part * eg::KPP;
    const double* pv = L.xv + part * eg::KPP;
    d2_t av[eg::KPP];
    double xk[eg::KPP];
#pragma unroll
    for (int kk = 0; kk < eg::KPP; kk++) {
      av[kk] = *reinterpret_cast<const d2_t*>(pa + 2 * kk);
      xk[kk] = pv[kk];
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int kk = 0; kk < eg::KPP; kk++) {
      s0 = __builtin_fma(av[kk][0], xk[kk], s0);
      s1 = __builtin_fma(av[kk][1], xk[kk], s1);
    }
    d2_t o; o[0] = s0; o[1] = s1;
    *reinterpret_cast<d2_t*>(act ? L.pv + part * g::P + 2 * ip : trash) = o;
  }
}

template <int NB, int NE, bool FWD>
__device__ __forceinline__ double matvec_sum(const Lds<NB>& L, int i) {
  using g = Geo<NB>;
  using eg = EGeo<NB, NE>;
  constexpr int NP = FWD ? cmin(eg::NPARTF, cdiv(g::RP, eg::RPP)) : cmin(eg::NPARTB, cdiv(g::P, eg::KPP));   // parts holding a sum
  double s = L.pv[i];
#pragma unroll
  for (int q = 1; q < NP; q++) s += L.pv[q * g::P + i];
  return s;
}

// ---- the E role: the Runge-Kutta state of a problem, its vector recursion, its HBM stores --------------------------------
// Forward and backward are the same arithmetic: with V = what the P waves leave in W (product minus half the forcing term),
// the stage slope of the matrix is f = -(V + V^T) in BOTH directions (backward: Psi_{t-1} = Psi_t - dt (-G + W' + W'^T) with
// the G/2 already inside V), and the slope of the vector is c - A.v with c = b (forward) or dEsde_dm (backward); the
// backward sweep adds the jumps after the step (euler.py:139-149).  Step i of a sweep starts at grid point tidx(i) and
// ends at tidx(i + 1), tidx(i) = i forward, Np - 1 - i backward.
template <int NIT>
struct EState {
  d2_t xk[NIT], acc[NIT];      // S_k / Psi_t of this thread's items; RK4 / Heun slope sum
  double vk, v1, v2, v3;       // vector entry and its RK slopes
  double c0, c1, c2;           // vector forcing at the step's start / end point and at the next step's end point
  double jm, jm_next;          // vector jump behind this / the next step (backward)
  int n_obs_cur, n_obs_next;   // observation index behind this / the next step, or -1 (backward, sparse jumps)
  const double* cin;           // b (forward) / dEsde_dm (backward) of the problem, [Np][D]
  double* vout;                // m / lam, [Np][D]
  double* mout;                // S / Psi, [Np][D][D]
  int prob;
};

template <int METHOD, bool FWD, int NB, int NE, bool DENSEJ>
struct ERole {
  using g = Geo<NB>;
  static constexpr int NIT = EGeo<NB, NE>::NIT, NS = n_stages<METHOD>();
  using State = EState<NIT>;
  const OdeArgs& a;
  const ETab<NB, NE>& T;
  double* SIG;
  const int te;
  const bool vl;
  const unsigned D8, te8;

  __device__ __forceinline__ ERole(const OdeArgs& a_, const ETab<NB, NE>& T_, double* sig, int te_)
      : a(a_), T(T_), SIG(sig), te(te_), vl(te_ < a_.D), D8(8u * (unsigned)a_.D), te8(8u * (unsigned)te_) {}
  __device__ __forceinline__ bool item(int q) const { return ((T.mask >> q) & 1u) != 0u; }
  __device__ __forceinline__ bool row2(int q) const { return ((T.mask >> (8 + q)) & 1u) != 0u; }
  __device__ __forceinline__ int tidx(int i) const { return FWD ? i : a.Np - 1 - i; }
  __device__ __forceinline__ size_t mat(int t) const { return (size_t)t * a.D * a.D; }
  __device__ __forceinline__ size_t vec(int t) const { return (size_t)t * a.D; }

  __device__ __forceinline__ void store_state(const State& S, int t) const {
    double* so = S.mout + mat(t);
#pragma unroll
    for (int q = 0; q < NIT; q++) {
      if (item(q)) {
        stg(so, T.gofs[q], S.xk[q][0]);
        if (row2(q)) stg(so, T.gofs[q] + D8, S.xk[q][1]);
      }
    }
    if (vl) stg(S.vout + vec(t), te8, S.vk);
  }

  __device__ __forceinline__ double jump_vector(const State& S, int t, int n_obs) const {
    if (!FWD) {
      if (DENSEJ) { if (vl) return ldg(a.jm_dense + ((size_t)S.prob * a.Np + t) * a.D, te8); }
      else if (vl && n_obs >= 0) return ldg(a.jm_sparse + ((size_t)S.prob * a.n_obs + n_obs) * a.D, te8);
    }
    return 0.0;
  }

  // state at the first grid point of the sweep, published; forcing of the first step
  __device__ __forceinline__ void prologue(State& S, int prob, const Lds<NB>& L, bool write_sig) const {
    const int Np = a.Np, n_steps = Np - 1;
    S.prob = prob;
    S.cin = FWD ? a.b + (size_t)prob * a.strideB : a.dEm + (size_t)prob * Np * a.D;
    S.vout = (FWD ? a.m : a.lam) + (size_t)prob * Np * a.D;
    S.mout = (FWD ? a.S : a.psi) + (size_t)prob * Np * a.D * a.D;
    S.vk = S.v1 = S.v2 = S.v3 = 0.0; S.jm = S.jm_next = 0.0; S.n_obs_cur = S.n_obs_next = -1; S.c2 = 0.0;
#pragma unroll
    for (int q = 0; q < NIT; q++) {
      S.xk[q] = d2_t{0.0, 0.0}; S.acc[q] = d2_t{0.0, 0.0};
      if (FWD && item(q)) {
        S.xk[q][0] = ldg(a.S0, T.gofs[q]);
        S.xk[q][1] = row2(q) ? ldg(a.S0, T.gofs[q] + D8) : 0.0;
        *reinterpret_cast<d2_t*>(L.X + T.offX[q]) = S.xk[q];
      }
      if (!FWD && !DENSEJ && write_sig && item(q)) {     // the constant matrix jump, once per workgroup
        d2_t js{0.0, 0.0};
        if (a.js_const) { js[0] = ldg(a.js_const, T.gofs[q]); js[1] = row2(q) ? ldg(a.js_const, T.gofs[q] + D8) : 0.0; }
        *reinterpret_cast<d2_t*>(SIG + 2 * (te + q * NE)) = js;
      }
    }
    if (FWD && vl) { S.vk = ldg(a.m0, te8); L.xv[te] = S.vk; }
    store_state(S, tidx(0));
    S.c0 = vl ? ldg(S.cin + vec(tidx(0)), te8) : 0.0;
    S.c1 = (vl && n_steps >= 1) ? ldg(S.cin + vec(tidx(1)), te8) : 0.0;
    if (!FWD && n_steps >= 1) {
      S.n_obs_cur = (!DENSEJ && a.obs_idx) ? ldu(a.obs_idx, tidx(1)) : -1;
      S.jm = jump_vector(S, tidx(1), S.n_obs_cur);
    }
#pragma unroll
    for (int q = 0; q < NIT; q++) settle(S.xk[q]);
    settle(S.c0); settle(S.c1); settle(S.vk); settle(S.jm);
  }

  // prefetches for the NEXT step (they arrive while this one runs)
  __device__ __forceinline__ void head(State& S, int i) const {
    const int n_steps = a.Np - 1;
    S.c2 = (vl && i + 2 <= n_steps) ? ldg(S.cin + vec(tidx(i + 2)), te8) : 0.0;
    if (!FWD) {
      S.n_obs_next = (!DENSEJ && a.obs_idx && i + 2 <= n_steps) ? ldu(a.obs_idx, tidx(i + 2)) : -1;
      S.jm_next = (i + 2 <= n_steps) ? jump_vector(S, tidx(i + 2), S.n_obs_next) : 0.0;
    }
  }
  __device__ __forceinline__ void tail(State& S) const {
    S.c0 = S.c1; S.c1 = S.c2; S.jm = S.jm_next; S.n_obs_cur = S.n_obs_next;
  }

  // beside the product of stage j of step i of this problem: partial sums of the vector's inner product, HBM stores
  __device__ __forceinline__ void chores(const State& S, const Lds<NB>& L, int j, int i) const {
    const int op = stage_op<METHOD, FWD>(j, false, i);
    matvec_partials<NB, NE, FWD>(op == OP_M ? L.M : L.R, L, te);
    if (j == 0 && i > 0) store_state(S, tidx(i));
  }

  // the element-wise stage j of step i: next stage state published, vector entry published
  __device__ __forceinline__ void elem(State& S, const Lds<NB>& L, int j, int i) const {
#pragma clang fp contract(fast)
    constexpr double sixth = 1.0 / 6.0;
    const double dt = a.dt, h = 0.5 * a.dt;
    const bool last = (j == NS - 1);
    double vs = 0.0;
    if (vl) vs = matvec_sum<NB, NE, FWD>(L, te);
    // In batches of up to four items: every LDS read of the batch first, then the arithmetic, then the publish.  No
    // branches (threads without an item read element (0, 0) and publish into their trash unit); one LDS round trip per
    // batch; the scheduling barrier keeps the reads of the next batch (and their registers) behind this one.
#pragma unroll
    for (int q0 = 0; q0 < NIT; q0 += 4) {
      d2_t wv[4], wt[4], js[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = q0 + u;
        if (q < NIT) {
          wv[u][0] = L.W[T.offW[q]]; wv[u][1] = L.W[T.offW[q] + g::LDW];
          wt[u] = *reinterpret_cast<const d2_t*>(L.W + T.offWt[q]);
          js[u] = d2_t{0.0, 0.0};
          if (!FWD && last) {      // matrix jump behind the step
            if (DENSEJ) {
              if (item(q)) {
                const double* jp = a.js_dense + ((size_t)S.prob * a.Np + tidx(i + 1)) * a.D * a.D;
                js[u][0] = ldg(jp, T.gofs[q]); js[u][1] = row2(q) ? ldg(jp, T.gofs[q] + D8) : 0.0;
              }
            } else if (S.n_obs_cur >= 0) {
              js[u] = *reinterpret_cast<const d2_t*>(SIG + 2 * (te + q * NE));
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = q0 + u;
        if (q < NIT) {
          d2_t xn;
          const d2_t f = -wv[u] - wt[u];
          if (METHOD == VGPA_ODE_EULER) {
            S.xk[q] = S.xk[q] + f * dt + js[u]; xn = S.xk[q];
          } else if (METHOD == VGPA_ODE_HEUN) {
            if (j == 0) { S.acc[q] = f; xn = S.xk[q] + f * dt; }
            else { S.xk[q] = S.xk[q] + h * (S.acc[q] + f) + js[u]; xn = S.xk[q]; }
          } else if (METHOD == VGPA_ODE_RK2) {
            if (j == 0) xn = S.xk[q] + h * f;
            else { S.xk[q] = S.xk[q] + dt * f + js[u]; xn = S.xk[q]; }
          } else {                                      // acc = k1 + 2 k2 + 2 k3 + k4, summed as the stages come
            if (j == 0) { S.acc[q] = f; xn = S.xk[q] + h * f; }
            else if (j == 1) { S.acc[q] = S.acc[q] + 2.0 * f; xn = S.xk[q] + h * f; }
            else if (j == 2) { S.acc[q] = S.acc[q] + 2.0 * f; xn = S.xk[q] + dt * f; }
            else { S.xk[q] = S.xk[q] + (dt * (S.acc[q] + f)) * sixth + js[u]; xn = S.xk[q]; }
          }
          *reinterpret_cast<d2_t*>(L.X + T.offX[q]) = xn;
        }
      }
      if (q0 + 4 < NIT) __builtin_amdgcn_sched_barrier(0);
    }
    if (vl) {
      const double cmid = 0.5 * (S.c0 + S.c1);
      double vn;
      if (METHOD == VGPA_ODE_EULER) { S.vk = S.vk + (S.c0 - vs) * dt + S.jm; vn = S.vk; }
      else if (METHOD == VGPA_ODE_HEUN) {
        if (j == 0) { S.v1 = S.c0 - vs; vn = S.vk + S.v1 * dt; }
        else { S.vk = S.vk + h * (S.v1 + (S.c1 - vs)) + S.jm; vn = S.vk; }
      } else if (METHOD == VGPA_ODE_RK2) {
        if (j == 0) vn = S.vk + h * (S.c0 - vs);
        else { S.vk = S.vk + dt * (cmid - vs) + S.jm; vn = S.vk; }
      } else {
        if (j == 0) { S.v1 = S.c0 - vs; vn = S.vk + h * S.v1; }
        else if (j == 1) { S.v2 = cmid - vs; vn = S.vk + h * S.v2; }
        else if (j == 2) { S.v3 = cmid - vs; vn = S.vk + dt * S.v3; }
        else { S.vk = S.vk + (dt * (S.v1 + 2.0 * (S.v2 + S.v3) + (S.c1 - vs))) * sixth + S.jm; vn = S.vk; }
      }
      L.xv[te] = vn;
    }
  }
};

// One workgroup integrates one problem: the phases alternate product / element-wise stage, one workgroup barrier each.
template <int METHOD, bool FWD, int NB, bool DENSEJ>
__device__ __forceinline__ void e_role(const OdeArgs& a, int prob, const Lds<NB>& L, double* __restrict__ SIG, int te) {
  using Role = ERole<METHOD, FWD, NB, kNE, DENSEJ>;
  constexpr int NS = Role::NS;
  const int n_steps = a.Np - 1;
  // The E waves issue few instructions, all on the critical path; the P wave of the same SIMD has MFMAs in flight when the
  // element-wise phase starts.  Without priority their fp64 VALU work would wait for the matrix pipe.
  __builtin_amdgcn_s_setprio(3);
  ETab<NB, kNE> T;
  build_etab<NB, kNE>(a.D, te, T);
  const Role R(a, T, SIG, te);
  typename Role::State S;
  __syncthreads();                                         // LDS zero-filled
  R.prologue(S, prob, L, true);
  __syncthreads();                                         // prologue published
  VGPA_STAMP_DECL;
  for (int k = 0; k < n_steps; k++) {
    R.head(S, k);
#pragma unroll
    for (int j = 0; j < NS; j++) {
      R.chores(S, L, j, k);              // beside the product
      VGPA_STAMP(1, 0);
      __syncthreads();
      VGPA_STAMP(1, 1);
      R.elem(S, L, j, k);
      VGPA_STAMP(1, 2);
      __syncthreads();
      VGPA_STAMP(1, 4);
    }
    R.tail(S);
  }
  if (n_steps > 0) R.store_state(S, R.tidx(n_steps));
}

// =================================================================================================================
// Workgroup = 8 waves = 4 P waves + 4 E waves, one problem.
template <int METHOD, bool FWD, int NB, bool DENSEJ>
__global__ void __launch_bounds__(64 * kNPW + kNE) k_ode_pe(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = Geo<NB>;
  constexpr int NT = 64 * kNPW + kNE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prob = (int)blockIdx.x;
  Lds<NB> L;
  L.carve(smem);
  double* SIG = smem + g::PROB;
  for (int i = tid; i < (int)g::LDS_DOUBLES; i += NT) smem[i] = 0.0;
  if (wave < kNPW) p_role<METHOD, FWD, NB>(a, prob, L, wave, lane);
  else e_role<METHOD, FWD, NB, DENSEJ>(a, prob, L, SIG, tid - 64 * kNPW);
}

template <int METHOD, bool FWD, int NB>
hipError_t launch_nb(const OdeArgs& a, hipStream_t st) {
  constexpr size_t lds = Geo<NB>::LDS_DOUBLES * sizeof(double);
  static_assert(lds <= 160 * 1024, "LDS budget");
  const bool dense = !FWD && a.js_dense;
  auto kern = dense ? k_ode_pe<METHOD, FWD, NB, true> : k_ode_pe<METHOD, FWD, NB, false>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(a.batch), dim3(64 * kNPW + kNE), lds, st, a);
  return hipGetLastError();
}

}  // namespace mfma

// One instantiation set per stepper (defined in ode_mfma_m<METHOD>.hip).
template <int METHOD> bool mfma_method_supported(int nb);
template <int METHOD> hipError_t mfma_method_launch(bool fwd, const OdeArgs& a, hipStream_t st);

}  // namespace vgpa
