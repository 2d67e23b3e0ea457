// E_sde integrand, <f>, <df/dx> and dE_sde/dm, dE_sde/dS per grid point.
//
//   OU  : src/dynamics/ornstein_uhlenbeck.py:165-232      (one thread per grid point)
//   DW  : src/dynamics/double_well.py:169-260 with the Gaussian moments of
//         src/var_bayes/gaussian_moments.py:43-183        (one thread per grid point; quirk Q7)
//   L63 : src/dynamics/lorenz_63.py:237-346, 348-568      (one thread per grid point, closed form)
//   L96 : src/dynamics/lorenz_96.py:316-438 with ut_approx (src/numerics/utilities.py:239-310) and
//         grad_Esde_dm_ds (src/var_bayes/variational.py:339-400): one workgroup per grid point,
//         Cholesky / triangular inverse / A.L / scaled SYRK in LDS, using the identities of
//         SURVEY.md s.8a ("Algebra the kernels may exploit"); the flat np.roll of the sigma-point
//         matrix (quirk Q1) is reproduced exactly.
#include "vgpa_internal.h"
#include "energy_small.h"
#include "chol_wave.h"

#define VGPA_L96_NOK 0
#define VGPA_L96_NOPANEL 0

namespace vgpa {
namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------------------
//  1-D models and Lorenz-63: one thread per grid point; the arithmetic lives in energy_small.h (shared with the fused
//  lane-per-problem sweep of ode_small.hip, which evaluates the same terms in registers instead of reading them back)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) k_energy_1d(EnergyArgs a) {
  const int t = blockIdx.x * NT + threadIdx.x;
  const int prob = blockIdx.y;
  if (t >= a.Np) return;
  const size_t o = (size_t)prob * a.Np + t;
  const double la = a.A[(size_t)prob * a.strideA + t], ob = a.b[(size_t)prob * a.strideB + t], m = a.m[o], s = a.S[o];
  Energy1d r;
  if (a.model == VGPA_MODEL_OU) energy_1d<VGPA_MODEL_OU>(a.theta[0], a.sigma1, la, ob, m, s, r);
  else energy_1d<VGPA_MODEL_DW>(a.theta[0], a.sigma1, la, ob, m, s, r);
  a.e_t[o] = r.e_t;
  a.Ef[o] = r.ef;
  if (a.Edf) a.Edf[o] = r.edf;
  a.dEm[o] = r.dm;
  a.dEs[o] = r.ds;
  if (a.hyp) a.hyp[o] = r.hyp;
}

__global__ void __launch_bounds__(64) k_energy_l63(EnergyArgs a) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int prob = blockIdx.y;
  if (t >= a.Np) return;
  const size_t o = (size_t)prob * a.Np + t;
  const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * 9;
  const double* bt = a.b + (size_t)prob * a.strideB + (size_t)t * 3;
  const double* mt = a.m + o * 3;
  const double* St = a.S + o * 9;
  double Av[9], bv[3], mv[3], Sv[9], isg[3];
#pragma unroll
  for (int e = 0; e < 9; e++) { Av[e] = At[e]; Sv[e] = St[e]; }
#pragma unroll
  for (int i = 0; i < 3; i++) { bv[i] = bt[i]; mv[i] = mt[i]; isg[i] = a.isg[i]; }
  EnergyL63 r;
  if (a.hyp) energy_l63<true>(a.theta, isg, Av, bv, mv, Sv, r); else energy_l63<false>(a.theta, isg, Av, bv, mv, Sv, r);
  a.e_t[o] = r.e_t;
  double* dm = a.dEm + o * 3;
  dm[0] = r.dm[0]; dm[1] = r.dm[1]; dm[2] = r.dm[2];
  double* ds = a.dEs + o * 9;
  ds[0] = r.ds[0]; ds[1] = r.ds[1]; ds[2] = r.ds[2];
  ds[3] = r.ds[1]; ds[4] = r.ds[3]; ds[5] = r.ds[4];
  ds[6] = r.ds[2]; ds[7] = r.ds[4]; ds[8] = r.ds[5];
  double* ef = a.Ef + o * 3;
  ef[0] = r.ef[0]; ef[1] = r.ef[1]; ef[2] = r.ef[2];
  if (a.hyp) {
    double* hp = a.hyp + o * 6;
#pragma unroll
    for (int i = 0; i < 6; i++) hp[i] = r.hyp[i];
  }
  if (a.Edf) {
    double* e = a.Edf + o * 9;
    const double vS = a.theta[0], vR = a.theta[1], vB = a.theta[2];
    e[0] = -vS; e[1] = vS; e[2] = 0.0;
    e[3] = vR - mv[2]; e[4] = -1.0; e[5] = -mv[0];
    e[6] = mv[1]; e[7] = mv[0]; e[8] = -vB;
  }
}

// ------------------------------------------------------------------------------------------------
//  Lorenz-96: ONE WAVE per grid point (64-thread workgroups, no workgroup barrier anywhere).
//
//  Per-wave LDS: Lm (c*S -> its lower Cholesky factor L, later q_k * L^-1[k][j]) and Gm (A, then A.L in place,
//  later L^-1), both [D][LD] with LD odd, plus a few vectors.  Phases and the lane mapping of each:
//    1. Cholesky, left-looking by columns            lane = row i        (cross-lane: pivot row reads, wave_sync)
//    2. A.m and G = A.L in place on A (ascending r)  lane = row i        (lane-private rows)
//    3. v_p = sum_i isg_i (l96_flat(chi)_pi + (A chi_p)_i - b_i)^2
//                                                    lane = SIGMA POINT p (ceil(M/64) passes); the sum
//       over i runs sequentially inside the lane with a sliding window over chi(p, i-2..i+1): no reduction.
//       The flat np.roll of the reference (quirk Q1) makes row p's neighbours at the row ends come from rows p-1, p+1.
//    4. L^-1 by forward substitution                 lane = column c     (lane-private columns)
//    5. dE/dm = (c/2) L^-T delta, dE/dS = (c/2) L^-T diag(q) L^-1        lane = column j
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of lane j (wave-uniform j) through v_readlane_b32: much cheaper than a ds_bpermute round trip
__device__ __forceinline__ double lane_value(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

using cholw::chol_panel_pivots;
using cholw::rsqrt_pos;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- wave-level fp64 MFMA product for the two GEMM-shaped phases --------------------------------------------------
// W[i][j] = sum_k Aop[k][i] * Bm[k][j] over the padded P x P problem (P = 4*NB), one wave, v_mfma_f64_4x4x4_4b_f64.
// Same unit scheme as the stepping kernels (ode_mfma_impl.h): a unit = one accumulator = four 4x4 output blocks;
// full units (I, q) cover block-row I x 16 columns, left-over units pack the remaining column blocks of G rows.
// Both callers have triangular structure that is exploited at compile time (whole 4x4x4 block products skipped):
//   MODE 0:  Bm[k][j] = 0 for k < j            (G = A.L, L lower triangular)
//   MODE 1:  additionally Aop[k][i] = 0 for k < i  (X^T diag(q) X with X = L^-1 lower triangular)
template <int NB>
struct WaveGemmGeo {
  static constexpr int NQ = NB / 4, REM = NB % 4, G = REM ? 4 / REM : 0;
  static constexpr int NLEFT = REM ? (NB + G - 1) / G : 0;
  static constexpr int NU = NB * NQ + NLEFT;
};

// TRI = 1: only the units that hold an element with row <= col (a symmetric result of which the caller keeps the upper triangle);
// TRI = 2: ... with row >= col (the lower triangle)
template <int NB, int MODE, int TRI = 0>
__device__ __forceinline__ void wave_gemm(const double* __restrict__ Aop, const double* __restrict__ Bm, int LD,
                                          double (&acc)[WaveGemmGeo<NB>::NU], const double* __restrict__ bscale = nullptr) {
  using g = WaveGemmGeo<NB>;
  const int l = threadIdx.x & 63, r4 = l >> 4, c4 = l & 3, b = (l >> 2) & 3;
  constexpr int rem = g::REM ? g::REM : 1;
#pragma unroll
  for (int u = 0; u < g::NU; u++) acc[u] = 0.0;
  const double* pa = Aop + r4 * LD + c4;
  const double* pb = Bm + r4 * LD;
  const int colq = l & 15;                                   // column inside a full 16-column group
  const int coll = 4 * (4 * g::NQ + b % rem) + c4;           // column of the left-over group
  const int ileft = b / rem;                                 // block-row offset inside a left-over unit
#pragma unroll
  for (int kk = 0; kk < NB; kk++) {
    const double bs = bscale ? bscale[4 * kk + r4] : 1.0;      // B[k][j] <- bscale[k] * B[k][j] applied to the fragment
    double bq[g::NQ > 0 ? g::NQ : 1];
#pragma unroll
    for (int q = 0; q < g::NQ; q++)
      if (kk >= 4 * q) bq[q] = bscale ? bs * pb[kk * 4 * LD + 16 * q + colq] : pb[kk * 4 * LD + 16 * q + colq];
    double bl = 0.0;
    if (g::NLEFT > 0 && kk >= 4 * g::NQ) bl = bscale ? bs * pb[kk * 4 * LD + coll] : pb[kk * 4 * LD + coll];
#pragma unroll
    for (int I = 0; I < NB; I++) {
      if (MODE == 1 && kk < I) continue;
      bool any = false;
#pragma unroll
      for (int q = 0; q < g::NQ; q++) any = any || (kk >= 4 * q && !(TRI == 1 && 4 * q + 3 < I) && !(TRI == 2 && 4 * q > I));
      if (!any) continue;
      const double af = pa[kk * 4 * LD + 4 * I];
#pragma unroll
      for (int q = 0; q < g::NQ; q++)
        if (kk >= 4 * q && !(TRI == 1 && 4 * q + 3 < I) && !(TRI == 2 && 4 * q > I)) acc[q * NB + I] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bq[q], acc[q * NB + I], 0, 0, 0);
    }
    if (g::NLEFT > 0 && kk >= 4 * g::NQ) {
#pragma unroll
      for (int v = 0; v < g::NLEFT; v++) {
        if (MODE == 1 && kk < v * g::G) continue;
        if (TRI == 2 && v * g::G + g::G - 1 < 4 * g::NQ) continue;      // every row block of the unit lies above its column blocks
        int ib = v * g::G + ileft;
        ib = ib < NB ? ib : NB - 1;                          // spare block slots read valid memory; result unused
        const double af = pa[kk * 4 * LD + 4 * ib];
        acc[NB * g::NQ + v] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bl, acc[NB * g::NQ + v], 0, 0, 0);
      }
    }
  }
}

// (row, col) of the element this lane holds in accumulator u; ok = it is a real block slot
template <int NB>
__device__ __forceinline__ void wave_gemm_elem(int u, int& row, int& col, bool& ok) {
  using g = WaveGemmGeo<NB>;
  const int l = threadIdx.x & 63, r4 = l >> 4, c4 = l & 3, b = (l >> 2) & 3;
  constexpr int rem = g::REM ? g::REM : 1;
  int Ib, Jb;
  if (u < NB * g::NQ) {
    const int q = u / NB;
    Ib = u - q * NB; Jb = 4 * q + b; ok = true;
  } else {
    const int v = u - NB * g::NQ;
    Ib = v * g::G + b / rem; Jb = 4 * g::NQ + b % rem;
    ok = (b < g::G * g::REM) && (Ib < NB);
  }
  row = 4 * Ib + r4; col = 4 * Jb + c4;
}

struct L96Lds {
  double *Lm, *Gm, *mv, *bv, *am, *sg, *vv, *dl, *qq, *rd;
};

__host__ __device__ inline int l96_dp(int D) { return 4 * ((D + 3) / 4); }          // padded to a multiple of 4
__host__ __device__ inline int l96_ld(int D) { return l96_dp(D) + 1; }               // odd leading dimension
__host__ __device__ inline size_t l96_lds_doubles(int D) {
  return (size_t)2 * l96_dp(D) * l96_ld(D) + 7 * (size_t)l96_dp(D) + (2 * D + 1) + 8 + 4 * (size_t)l96_dp(D);   // + xdiag
}

// Register blocking: every inner k-iteration below feeds FOUR independent fma chains from FIVE LDS reads (one
// lane-private operand + four wave-uniform ones), which both cuts LDS traffic (the binding resource of this
// kernel) and gives the scheduler independent work to hide the LDS latency behind.
template <int NB>
__global__ void __launch_bounds__(64) k_energy_l96(EnergyArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int D = a.D, M = 2 * D + 1;
  constexpr int Dp = 4 * NB, LD = Dp + 1;
  const int l = threadIdx.x;
  const long long wid = blockIdx.x;
  const int prob = (int)(wid / a.Np), t = (int)(wid - (long long)prob * a.Np);
  const size_t o = (size_t)prob * a.Np + t;
  L96Lds S;
  S.Lm = smem; S.Gm = S.Lm + Dp * LD; S.mv = S.Gm + Dp * LD; S.bv = S.mv + Dp; S.am = S.bv + Dp; S.sg = S.am + Dp;
  S.dl = S.sg + Dp; S.qq = S.dl + Dp; S.rd = S.qq + Dp; S.vv = S.rd + Dp;
  const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * D * D;
  const double* St = a.S + o * D * D;
  const double theta = a.theta[0];
  const double kappa = 1.05 * D, c = D + kappa;
  const bool act = l < D;                  // lane owns a real row / column
  const bool pad = l < Dp;                 // lane owns a row / column of the padded problem
  const int li = pad ? l : Dp - 1;         // clamped index so that idle lanes read valid memory

  // ---- stage c*S and A into LDS (coalesced); padding: identity for c*S, zero for A and the vectors
  for (int e = l; e < Dp * LD; e += 64) { S.Lm[e] = 0.0; S.Gm[e] = 0.0; }
  if (pad) { S.mv[l] = 0.0; S.bv[l] = 0.0; S.sg[l] = 0.0; S.am[l] = 0.0; S.dl[l] = 0.0; S.qq[l] = 0.0; S.rd[l] = 1.0; }
  wave_sync();
  {
    // all HBM loads of a chunk are issued before the first one is consumed (a lone wave cannot hide a serial chain
    // of ~2 us HBM round trips); e / D by multiplication with ceil(2^20 / D), exact for e < 4096, 4 <= D <= 64
    constexpr int EPL = (Dp * Dp + 63) / 64;
    const int DD = D * D;
    const unsigned magic = ((1u << 20) + (unsigned)D - 1u) / (unsigned)D;
#pragma unroll
    for (int q0 = 0; q0 < EPL; q0 += 9) {
      double sv[9], av[9];
#pragma unroll
      for (int u = 0; u < 9; u++) {
        const int e = l + 64 * (q0 + u);
        const bool in = (q0 + u < EPL) && (e < DD);
        sv[u] = in ? St[e] : 0.0;
        av[u] = in ? At[e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 9; u++) {
        const int e = l + 64 * (q0 + u);
        if ((q0 + u < EPL) && (e < DD)) {
          const int r = (int)(((unsigned)e * magic) >> 20), cc = e - r * D;
          S.Lm[r * LD + cc] = c * sv[u];
          S.Gm[cc * LD + r] = av[u];               // A^T: operand layout [k][i] of the MFMA product
        }
      }
    }
  }
  if (l >= D && pad) S.Lm[l * LD + l] = 1.0;
  if (act) { S.mv[l] = a.m[o * D + l]; S.bv[l] = a.b[(size_t)prob * a.strideB + (size_t)t * D + l]; S.sg[l] = a.isg[l]; }
  wave_sync();

  // ---- 1. Cholesky, left-looking in panels of four columns (numpy.linalg.cholesky reads the lower triangle); the
  //         strict upper triangle is zeroed on the way.  Per panel p the update U = L[4p:, :4p] . L[4p:4p+4, :4p]^T runs
  //         on the matrix cores (unit u = block-rows p + 4u + b; A-operand [i][k] = L[4I + i][4kk + k], B-operand
  //         [k][j] = L[4p + j][4kk + k]) and is subtracted from the panel in place; then the four pivots (lane = row).
  const int r4 = l >> 4, c4 = l & 3, b = (l >> 2) & 3;
  // (fully unrolled: NB is a template parameter, so every LDS offset below is an immediate and the loops cost no
  // scalar bookkeeping -- this kernel is bound by instructions issued per grid point.  A non-positive pivot makes the
  // rest of the factor NaN, which is harmless: the flag is checked after the loop.)
  bool bad = false;
  double myrd = 1.0;                       // 1 / L[l][l] (padding rows: 1), stored once after the loop
  const double* lrow_b = S.Lm + (4 * b + c4) * LD + r4;      // + 4 (p + 4u) LD: A-operand rows of block b
  double* lout_b = S.Lm + (4 * b + r4) * LD + c4;            // + 4 (p + 4u) LD + 4p: output position of block b
#pragma unroll
  for (int p = 0; p < NB; p++) {
    const int j0 = 4 * p;
    if (p > 0) {
      const double* brow = S.Lm + (j0 + c4) * LD + r4;
#pragma unroll
      for (int u = 0; p + 4 * u < NB; u++) {
        const bool full = p + 4 * u + 3 < NB;                // compile-time: all four block-rows of the unit exist
        const bool rowok = full || (p + 4 * u + b < NB);
        const double* arow = full ? lrow_b + 4 * (p + 4 * u) * LD
                                  : S.Lm + (4 * (rowok ? p + 4 * u + b : NB - 1) + c4) * LD + r4;
        double uacc = 0.0;
#pragma unroll
        for (int kk = 0; kk < p; kk++) uacc = __builtin_amdgcn_mfma_f64_4x4x4f64(arow[4 * kk], brow[4 * kk], uacc, 0, 0, 0);
        if (rowok) lout_b[4 * (p + 4 * u) * LD + j0] -= uacc;  // D[i = r4][j = c4] of block b
      }
      wave_sync();
    }
    chol_panel_pivots<LD>(S.Lm, myrd, j0, l, li, pad);       // 1/sqrt and sqrt to ~1 ulp, no fp64 divide
    wave_sync();
  }
  if (pad) S.rd[l] = myrd;
  wave_sync();
  bad = __any(!(myrd > 0.0 && myrd < __builtin_inf()));
  if (bad) {
    // S_t is not positive definite: the reference raises LinAlgError from chol_inv(S_t) (variational.py:380)
    // after its diagonal fallback (utilities.py:279); report and stop.
    if (l == 0) atomicOr(a.status + prob, 1);
    return;
  }

  // inverses of the 4x4 diagonal blocks of L (lane I < NB), used by the blocked forward substitution of phase 4
  double* xdiag = S.vv + (2 * D + 1) + 7;
  if (l < NB) {
    const double* tb = S.Lm + (4 * l) * LD + 4 * l;
    const double x00 = S.rd[4 * l], x11 = S.rd[4 * l + 1], x22 = S.rd[4 * l + 2], x33 = S.rd[4 * l + 3];
    const double t10 = tb[LD], t20 = tb[2 * LD], t21 = tb[2 * LD + 1], t30 = tb[3 * LD], t31 = tb[3 * LD + 1], t32 = tb[3 * LD + 2];
    const double x10 = -(t10 * x00) * x11;
    const double x21 = -(t21 * x11) * x22;
    const double x32 = -(t32 * x22) * x33;
    const double x20 = -(t20 * x00 + t21 * x10) * x22;
    const double x31 = -(t31 * x11 + t32 * x21) * x33;
    const double x30 = -(t30 * x00 + t31 * x10 + t32 * x20) * x33;
    double* xo = xdiag + 16 * l;
    xo[0] = x00; xo[1] = 0.0; xo[2] = 0.0; xo[3] = 0.0;
    xo[4] = x10; xo[5] = x11; xo[6] = 0.0; xo[7] = 0.0;
    xo[8] = x20; xo[9] = x21; xo[10] = x22; xo[11] = 0.0;
    xo[12] = x30; xo[13] = x31; xo[14] = x32; xo[15] = x33;
  }
  // ---- 2. A.m (lane = row i, A^T rows are contiguous over i) ; G = A.L on the matrix cores, then Gm <- G ([i][r])
  {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < Dp; k++) s = __builtin_fma(S.Gm[k * LD + li], S.mv[k], s);   // padding rows of A^T and of m are zero
    if (act) { S.am[l] = s; if (a.Am) a.Am[o * D + l] = s; }
    double acc[WaveGemmGeo<NB>::NU];
    wave_gemm<NB, 0>(S.Gm, S.Lm, LD, acc);
    wave_sync();                                 // every lane is done reading A^T before G overwrites it
#pragma unroll
    for (int u = 0; u < WaveGemmGeo<NB>::NU; u++) {
      int row, col; bool ok;
      wave_gemm_elem<NB>(u, row, col, ok);
      if (ok) S.Gm[row * LD + col] = acc[u];
    }
  }
  wave_sync();

  // ---- 3. v_p.  Lane j < D evaluates BOTH sigma points of column j (p = 1+j: m + L[:,j] and p = 1+D+j: m - L[:,j]) in
  //         one sequential pass over i -- they share every LDS operand; the mean point p = 0 is evaluated in parallel
  //         over i.  chi(p, i) = m_i + sgn_p L[i][col_p]; the flat np.roll of the reference (quirk Q1) makes the
  //         neighbours at the row ends come from the sigma points p-1 and p+1.
  double vplus = 0.0, vminus = 0.0, v0 = 0.0, r0sq = 0.0;
  const bool hyper = a.hyp != nullptr;
  {
    auto col_of = [&](int q) { return q == 0 ? 0 : (q <= D ? q - 1 : q - 1 - D); };
    auto sgn_of = [&](int q) { return q == 0 ? 0.0 : (q <= D ? 1.0 : -1.0); };
    auto chi = [&](int q, int i) { return S.mv[i] + sgn_of(q) * S.Lm[i * LD + col_of(q)]; };
    const int jc = act ? l : 0;
    const int pP = 1 + jc, pM = 1 + D + jc;
    const int pPm = pP - 1, pPp = pP + 1;                         // 0 <= pP-1, pP+1 <= D+1 <= M-1
    const int pMm = pM - 1, pMp = wrap(pM + 1, M);
    // windows over the flat index w = p*D + i: xm2 = X(w-2), xm1 = X(w-1), x0 = X(w), x1 = X(w+1)
    double am2 = chi(pPm, D - 2), am1 = chi(pPm, D - 1), a0 = chi(pP, 0), a1 = chi(pP, 1);
    double bm2 = chi(pMm, D - 2), bm1 = chi(pMm, D - 1), b0 = chi(pM, 0), b1 = chi(pM, 1);
    const double* lcol = S.Lm + jc;
    const double* gcol = S.Gm + jc;
#pragma unroll
    for (int i = 0; i < Dp; i++) {                               // unrolled: immediate LDS offsets, no window moves
      if (i >= D) break;
      const double gi = gcol[i * LD], ami = S.am[i], bvi = S.bv[i], sgi = S.sg[i];
      const double ra = ((a1 - am2) * am1 - a0 + theta) + (ami + gi) - bvi;
      const double rb = ((b1 - bm2) * bm1 - b0 + theta) + (ami - gi) - bvi;
      vplus = __builtin_fma(sgi, ra * ra, vplus);
      vminus = __builtin_fma(sgi, rb * rb, vminus);
      if (hyper && act) S.Gm[i * LD + jc] = ra * ra + rb * rb;      // G[i][jc] has just been consumed by this lane
      am2 = am1; am1 = a0; a0 = a1;
      bm2 = bm1; bm1 = b0; b0 = b1;
      const int in = i + 2;
      if (in < D) {
        const double mi = S.mv[in], li2 = lcol[in * LD];
        a1 = mi + li2; b1 = mi - li2;
      } else {
        a1 = chi(pPp, in - D); b1 = chi(pMp, in - D);
      }
    }
    // the mean point: lane i holds term i of the sum
    if (act) {
      const int i = l;
      const double xm2 = (i >= 2) ? S.mv[i - 2] : chi(M - 1, D - 2 + i);
      const double xm1 = (i >= 1) ? S.mv[i - 1] : chi(M - 1, D - 1);
      const double x1 = (i + 1 < D) ? S.mv[i + 1] : chi(1, 0);
      const double r0 = ((x1 - xm2) * xm1 - S.mv[i] + theta) + S.am[i] - S.bv[i];
      r0sq = r0 * r0;
      v0 = S.sg[i] * r0sq;
    }
    v0 = wave_sum(v0);
  }
  const double w0 = kappa / c, w1 = 1.0 / (2.0 * c);
  if (hyper) {
    // dEsde_dth(t) = <f> + A m - b (lorenz_96.py:421), dEsde_dSig(t) = m_bar = UT mean of (f-g)^2 per component (:424):
    // lane i sums row i of the squared residuals left in Gm
    wave_sync();
    if (act) {
      double sm = 0.0;
      for (int j = 0; j < D; j++) sm += S.Gm[l * LD + j];
      const int i = l, ip1 = wrap(i + 1, D), im1 = wrap(i - 1, D), im2 = wrap(i - 2, D);
      const double efi = (St[ip1 * D + im1] - St[im2 * D + im1]) + (S.mv[ip1] - S.mv[im2]) * S.mv[im1] - S.mv[i] + theta;
      double* hp = a.hyp + o * 2 * D;
      hp[i] = efi + S.am[i] - S.bv[i];
      hp[D + i] = w0 * r0sq + w1 * sm;
    }
  }
  const double e_part = act ? (vplus + vminus) : 0.0;
  const double e_t = 0.5 * (w0 * v0 + w1 * wave_sum(e_part));
  if (act) {
    S.dl[l] = w1 * (vplus - vminus);
    S.qq[l] = 0.5 * c * (w1 * (vplus + vminus)) - e_t;
  }
  if (l == 0) a.e_t[o] = e_t;
  wave_sync();

  // ---- 4. X = L^-1 into Gm by blocked forward substitution on the matrix cores: for block-row I and the unit of column
  //         blocks J_b = 4u + b:  T_b = sum_{K < I} L[I][K] . X[K][J_b] accumulates in the MFMA result register, which is
  //         already laid out as the B-operand of the second product X[I][J_b] = -inv(L[I][I]) . T_b.
  const double* l4_a = S.Lm + c4 * LD + r4;                  // + 4 I LD + 4 K : A-operand L[4I + c4][4K + r4]
  const double* xd_a = xdiag + 4 * c4 + r4;                  // + 16 I
  const double* xd_d = xdiag + 4 * r4 + c4;
#pragma unroll
  for (int I = 0; I < NB; I++) {
    const double* arow = l4_a + 4 * I * LD;                  // A-operand [i][k] = L[4I + i][4K + k]
    const double xd = xd_a[16 * I];                          // A-operand [i][k] = inv(L[I][I])[i][k]
    const double xdd = xd_d[16 * I];                         // the diagonal block itself in D layout
#pragma unroll
    for (int u = 0; 4 * u < NB; u++) {
      const int Jb = 4 * u + b;
      const bool colok = (4 * u + 3 < NB) || (Jb < NB);
      double* xcol = S.Gm + r4 * LD + 4 * (colok ? Jb : NB - 1) + c4;
      double xo = 0.0;
      if (4 * u <= I) {                                      // compile-time: some block of the unit is on / below the diagonal
        double tacc = 0.0;
#pragma unroll
        for (int K = 4 * u; K < I; K++) tacc = __builtin_amdgcn_mfma_f64_4x4x4f64(arow[4 * K], xcol[4 * K * LD], tacc, 0, 0, 0);
        const double prod = __builtin_amdgcn_mfma_f64_4x4x4f64(xd, tacc, 0.0, 0, 0, 0);
        xo = (Jb < I) ? -prod : ((Jb == I) ? xdd : 0.0);
      }
      if (colok) xcol[4 * I * LD] = xo;                      // X[4I + r4][4 J_b + c4]
    }
    wave_sync();
  }

  // ---- 5. dE/dm = (c/2) X^T delta ; dE/dS = (c/2) X^T diag(q) X
  {
    double s = 0.0;
    for (int k = 0; k < Dp; k++) s = __builtin_fma(S.Gm[k * LD + li], S.dl[k], s);   // X[k][l] = 0 for k < l
    if (act) a.dEm[o * D + l] = 0.5 * c * s;
  }
  double* ds = a.dEs + o * D * D;
  {
    double acc[WaveGemmGeo<NB>::NU];
    wave_gemm<NB, 1>(S.Gm, S.Gm, LD, acc, S.qq);   // X^T (diag(q) X): q applied to the B fragments on the fly
#pragma unroll
    for (int u = 0; u < WaveGemmGeo<NB>::NU; u++) {
      int row, col; bool ok;
      wave_gemm_elem<NB>(u, row, col, ok);
      if (ok && row < D && col < D && (row <= col || !a.ds_upper)) ds[row * D + col] = 0.5 * c * acc[u];
    }
  }

  // ---- <f> (E96_drift, lorenz_96.py:440-462) and optionally dense <df/dx> (E96_drift_dx, :35-83)
  if (act) {
    const int i = l, ip1 = wrap(i + 1, D), im1 = wrap(i - 1, D), im2 = wrap(i - 2, D);
    const double cxx = St[ip1 * D + im1] - St[im2 * D + im1];
    a.Ef[o * D + i] = cxx + (S.mv[ip1] - S.mv[im2]) * S.mv[im1] - S.mv[i] + theta;
  }
  if (a.Edf) {
    double* ed = a.Edf + o * D * D;
    for (int e = l; e < D * D; e += 64) {
      const int k = e / D, j = e - k * D;
      const int kp1 = wrap(k + 1, D), km1 = wrap(k - 1, D), km2 = wrap(k - 2, D);
      // same assignment order as the reference: later assignments win when indices coincide
      double v = 0.0;
      if (j == k) v = -1.0;
      if (j == kp1) v = S.mv[km1];
      if (j == km2) v = -S.mv[km1];
      if (j == km1) v = S.mv[kp1] - S.mv[km2];
      ed[e] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
//  Lorenz-96, one wave per grid point with ONE matrix in LDS -- the shipped kernel whenever the hyper-parameter
//  integrands are not asked for.  k_energy_l96 above is bound by the number of waves a CU can hold (measured: 5 / 4 / 3
//  / 2 resident waves -> 6.35 / 7.7 / 9.9 / 14.0 ms), and that number is set by its two LDS matrices.  Here
//    * A never goes through LDS: every lane loads its MFMA A-operand fragments A[16u + (l & 15)][4K + (l >> 4)] straight
//      from HBM at kernel entry (they arrive while the Cholesky factorisation runs),
//    * A.m and G = A.L use those fragments with the blocks of L as the shared operand (I-major units), so G stays in
//      the accumulators: lane (b, r4, c4) holds G[16u + 4b + r4][4J + c4],
//    * the sigma-point residuals are evaluated in that accumulator layout (four LDS reads of L per element, rows
//      2 .. D-2; the three rows whose flat-roll neighbours belong to other sigma points -- 0, 1, D-1 (quirk Q1) -- are
//      done afterwards with lane = column from three parked rows of G), summed over the rows by one ones-MFMA (k = r4)
//      and four partial rows in LDS (b),
//    * L^-1 overwrites L in place (block-row I of L is dead once block-row I of the inverse is known).
//  LDS per grid point at D = 40: 17.9 KB instead of 30.5 KB -> 9 waves per CU instead of 5 (168 VGPRs: three per SIMD).
// ------------------------------------------------------------------------------------------------
// NB = 10 (37 <= D <= 40, the headline's size; round 5): the COMPACT layout -- L + FIVE vectors (m, b, A m, delta, q), 14.4 KB: ELEVEN
// waves per CU instead of nine.  Everything else lives where the matrix has room that nobody reads at that time:
//   * row D-1 of the parked G in the padding column of L (column Dp of the odd leading dimension),
//   * the partial sums of the residuals (one row of Dp for the plus, one for the minus points: the four block slots are summed by
//     row rotations in registers first) in rows 4 .. 7, columns 20 .. 39 -- strictly-upper blocks (1, 5 .. 9) that are exact zeros,
//     which only the residual loop (before) and nobody after it reads,
//   * the inverses of the ten diagonal blocks (phase 4) in the blocks (0, 4 .. 9) and (1, 5 .. 8): phase 4 reads blocks (K, J) with
//     J - K <= 3 only, the SYRK skips every block above the diagonal at compile time, dE/dm masks rows 0 .. 7,
//   * 1 / L_ii stays in the lanes' registers (four cross-lane reads build a diagonal block's inverse).
__host__ __device__ inline bool l96r_compact(int D) { return (D + 3) / 4 == 10; }
__host__ __device__ inline size_t l96r_lds_doubles(int D) {
  const size_t dp = l96_dp(D);
  if (l96r_compact(D)) return dp * l96_ld(D) + 5 * dp;
  return dp * l96_ld(D) + 6 * dp + 9 * dp;       // L + 6 vectors + scratch (4 + 4 partial rows, 1 row of G; later xdiag)
}
// one double rotated by R block slots inside every 16-lane row (v_mov_b32 row_ror:4R: lane l reads lane (l - 4R) mod 16 of its row)
template <int R>
__device__ __forceinline__ double row_ror_blocks(double v) {
  typedef int i2_t __attribute__((ext_vector_type(2)));
  i2_t x = __builtin_bit_cast(i2_t, v), y;
  y[0] = __builtin_amdgcn_mov_dpp(x[0], 0x120 + 4 * R, 0xf, 0xf, false);
  y[1] = __builtin_amdgcn_mov_dpp(x[1], 0x120 + 4 * R, 0xf, 0xf, false);
  return __builtin_bit_cast(double, y);
}

// row of flat index e of a packed lower triangle (D <= 64: e < 2080), as a table: the staging of a packed S_t looks its (row, column)
// up instead of computing a square root per element
struct TriRowTab { unsigned char r[2080]; };
constexpr TriRowTab make_tri_row_tab() {
  TriRowTab t{};
  int e = 0;
  for (int r = 0; r < 64; r++)
    for (int c = 0; c <= r; c++) t.r[e++] = (unsigned char)r;
  return t;
}
__device__ __constant__ TriRowTab kTriRow = make_tri_row_tab();

#ifdef VGPA_ENERGY_TRACE
// diagnostic build only: cycles per phase of k_energy_l96_r summed over all waves (tools/trace_energy.py)
__device__ unsigned long long g_energy_trace[16];
#define VGPA_TRACE_STAMP(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); if (l == 0) atomicAdd(&g_energy_trace[k], tn_ - tprev_); tprev_ = tn_; } while (0)
#else
#define VGPA_TRACE_STAMP(k) do { } while (0)
#endif

// TPW > 1 (packed S_t only): the wave takes TPW consecutive grid points of one problem and requests the next point's S_t (13
// registers of packed triangle at D = 40) and vector entries before the triangular inverse of the current one -- the phases behind
// that point need few registers, and the wave's entry and the memory round trip of its operands, 23 % of a one-point wave's time
// (profiles/r03_energy_kernel_phase_trace.json), run under them.
// (EXPERIMENT, off: -DVGPA_ENERGY_TPW=4 builds the persistent variant.  Measured 11.2 ms against 5.5 ms per 512-problem launch: with a
//  loop around the body the register allocator spills ~90 values (352 bytes of scratch per lane at the 168-register budget of three
//  waves per SIMD) wherever the request for the next point is placed -- EXPERIMENTS.md s.10.)
#ifndef VGPA_ENERGY_TPW
#define VGPA_ENERGY_TPW 1
#endif
constexpr int kEnergyTPW = VGPA_ENERGY_TPW;
template <int NB, int TPW = 1>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NB <= 10 ? 3 : 2, NB <= 10 ? 3 : 2))) k_energy_l96_r(EnergyArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int D = a.D, M = 2 * D + 1;
  constexpr int Dp = 4 * NB, LD = Dp + 1, NUU = (NB + 3) / 4;
  int l = threadIdx.x;
#ifdef VGPA_ENERGY_TRACE
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
  const long long wid = blockIdx.x;
  const int per = (a.Np + TPW - 1) / TPW;            // waves per problem
  const int prob = (int)(wid / per);
  int t = (int)(wid - (long long)prob * per) * TPW;
  const int t_last = t + TPW < a.Np ? t + TPW : a.Np;
  size_t o = (size_t)prob * a.Np + t;
  L96Lds S;
  // (Sigma^-1's diagonal is kept in registers, not in LDS; rows 0 and 1 of the parked G share the space of dl and qq,
  //  which are written after the boundary pass: 17.9 KB per grid point at D = 40, nine waves per CU)
  constexpr bool CMP = (NB == 10);   // the compact LDS layout (l96r_lds_doubles)
  S.Lm = smem; S.Gm = nullptr; S.mv = S.Lm + Dp * LD; S.bv = S.mv + Dp; S.am = S.bv + Dp; S.sg = nullptr;
  S.dl = S.am + Dp; S.qq = S.dl + Dp; S.rd = CMP ? nullptr : S.qq + Dp; S.vv = CMP ? nullptr : S.rd + Dp;
  double* pv = S.vv;                 // [4][Dp] partial sums (over the rows of block-slot b) of the plus points
  double* pw = CMP ? nullptr : pv + 4 * Dp;          // [4][Dp] ... of the minus points
  double* gb2 = CMP ? nullptr : pw + 4 * Dp;         // [Dp] row D-1 of G (rows 0 and 1: S.dl, S.qq)
  double* xdiag = pv;                // phase 4 (the partial sums are dead by then)
  // compact layout: element j of the parked row D-1 of G; of the plus / minus partial-sum rows; element (r, c) of diagonal-block inverse I
  auto cg2 = [&](int j) -> double* { return S.Lm + j * LD + Dp; };
  auto cps = [&](int minus, int j) -> double* { return S.Lm + (4 + 2 * minus + (j >= 20 ? 1 : 0)) * LD + (j >= 20 ? j : 20 + j); };
  auto cxd = [&](int I, int r, int c) -> double* { return S.Lm + ((I < 6 ? 0 : 4) + r) * LD + 4 * (I < 6 ? I + 4 : I - 1) + c; };
  const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * D * D;       // (advanced per grid point)
  const bool spk = a.s_packed != 0;                // S_t as its packed lower triangle (OdeArgs::s_packed): all the factorisation reads
  const int PK = tri_off(D);
  const double* St = a.S + o * (spk ? PK : D * D);
  static_assert(TPW >= 1, "grid points per wave");
  const double theta = a.theta[0];
  const double kappa = 1.05 * D, c = D + kappa;
  bool act = l < D;
  bool pad = l < Dp;
  int li = pad ? l : Dp - 1;
  int r4 = l >> 4, c4 = l & 3, b = (l >> 2) & 3;

  // ---- A-operand fragments of A, straight from HBM: unit u = block-rows 4u + b, fragment [i = c4][k = r4]
  // (the last unit's fragments are requested after the Cholesky instead: with them in flight the factorisation would not fit the
  //  168 registers of three waves per SIMD, and a spilled fragment is a load that is WAITED for at entry; they arrive during the
  //  first pass of phase 2)
#ifndef VGPA_ENERGY_EARLY_UNITS
#define VGPA_ENERGY_EARLY_UNITS 0
#endif
  constexpr int NUE = VGPA_ENERGY_EARLY_UNITS < NUU ? VGPA_ENERGY_EARLY_UNITS : NUU;
  double af[NUU][NB];
  const int klim = (D - r4 + 3) >> 2;                       // 4K + r4 < D  <=>  K < klim: nothing per K stays alive until the late loads
  auto load_af = [&](int u) {
    const int row = 16 * u + (l & 15);
    const double* ap = At + row * D + r4;
    const int kl = row < D ? klim : 0;
#pragma unroll
    for (int K = 0; K < NB; K++) af[u][K] = (K < kl) ? ap[4 * K] : 0.0;
  };
#pragma unroll
  for (int u = 0; u < NUE; u++) load_af(u);

  // the two entries of S_t that <f>_i needs at the very end (E96_drift): requested now -- at the end they would cost a
  // full memory round trip per wave
  double sxa = 0.0, sxb = 0.0, v_m = 0.0, v_b = 0.0, v_sg = 0.0;
  double nsxa = 0.0, nsxb = 0.0;          // packed path: the two entries of the grid point whose operands are in flight
  const int f_ip1 = wrap(l + 1, D), f_im1 = wrap(l - 1, D), f_im2 = wrap(l - 2, D);
  if (act) v_sg = a.isg[l];
  auto load_vectors = [&]() {
    if (act) {
      sxa = St[f_ip1 * D + f_im1];
      sxb = St[f_im2 * D + f_im1];
      v_m = a.m[o * D + l];
      v_b = a.b[(size_t)prob * a.strideB + (size_t)t * D + l];
    }
  };
  // packed lower triangle: flat, fully coalesced loads (lane = double e of the triangle: 13 requests at D = 40 instead of 40 row
  // loads); `request` asks for everything of grid point tn the wave reads from HBM except A
  constexpr int EPK = (Dp * (Dp + 1) / 2 + 63) / 64;
  double sv[EPK];
  int rr[TPW == 1 ? EPK : 1];              // rows of this lane's flat indices: looked up (kTriRow) beside the loads they belong to
  auto request = [&](int tn) {
    const size_t on = (size_t)prob * a.Np + tn;
    const double* Sn = a.S + on * PK;
#pragma unroll
    for (int q = 0; q < EPK; q++) {
      const int e = l + 64 * q;
      sv[q] = e < PK ? Sn[e] : 0.0;
      if (TPW == 1) rr[q] = kTriRow.r[e < PK ? e : 0];
    }
    if (act) {
      nsxa = Sn[tri_idx(f_ip1, f_im1)];
      nsxb = Sn[tri_idx(f_im2, f_im1)];
      v_m = a.m[on * D + l];
      v_b = a.b[(size_t)prob * a.strideB + (size_t)tn * D + l];
    }
  };
  if (spk) request(t);

  for (;;) {      // the grid points of this wave (one, unless TPW > 1)
  if (TPW > 1) {
    // The lane index is made opaque per grid point: everything a lane addresses derives from it, and with a loop around the body
    // the compiler otherwise hoists every per-lane LDS / HBM offset of every phase out of the loop -- ~300 registers of "invariants"
    // that a one-grid-point wave computes where it needs them (1.2 KB of scratch per lane before this line).
    asm volatile("" : "+v"(l));
    act = l < D; pad = l < Dp; li = pad ? l : Dp - 1;
    r4 = l >> 4; c4 = l & 3; b = (l >> 2) & 3;
  }
  // ---- stage c*S into LDS (coalesced); padding: identity.  Every HBM load of the wave (the vectors too) is requested
  //      before the first one is consumed: a lone wave pays each dependent round trip in full.  S_t's requests go first.
  if (spk) {
    // each value to its (row, column) of the LDS matrix; the strict upper triangle is never read before the factorisation's
    // pivots zero it (chol_panel_pivots)
    if (D != Dp) {
      for (int e = l; e < Dp * LD; e += 64) S.Lm[e] = 0.0;
      wave_sync();
    }
    sxa = nsxa; sxb = nsxb;
#pragma unroll
    for (int q = 0; q < EPK; q++) {
      const int e = l + 64 * q;
      if (e < PK) {
        const int r = (TPW > 1) ? tri_row(e) : rr[TPW == 1 ? q : 0];      // (persistent waves: arithmetic instead of 13 more registers)
        S.Lm[r * LD + (e - tri_off(r))] = c * sv[q];
      }
    }
  } else if (D == Dp) {
    // no padding (D = 40 of the headline): one row per load, lane = column -- every HBM and LDS offset is an immediate, no index
    // arithmetic per element; rows in flight per round trip are bounded by the registers (the A fragments are not in flight yet)
    constexpr int RCH = Dp <= 44 ? Dp : 32;
    const double* sp = St + li;
#pragma unroll
    for (int q0 = 0; q0 < Dp; q0 += RCH) {
      double sv[RCH];
#pragma unroll
      for (int u = 0; u < RCH; u++) sv[u] = (q0 + u < Dp) ? sp[(q0 + u) * Dp] : 0.0;   // (idle lanes re-read column Dp-1: no predicate)
      if (q0 == 0) load_vectors();
      if (pad) {
#pragma unroll
        for (int u = 0; u < RCH; u++)
          if (q0 + u < Dp) S.Lm[(q0 + u) * LD + l] = c * sv[u];
      }
    }
  } else {
    load_vectors();
    for (int e = l; e < Dp * LD; e += 64) S.Lm[e] = 0.0;
    wave_sync();
    constexpr int EPL = (Dp * Dp + 63) / 64;
    constexpr int CH = (NB <= 11) ? EPL : 13;                // loads in flight per round trip (registers)
    const int DD = D * D;
    const unsigned magic = ((1u << 20) + (unsigned)D - 1u) / (unsigned)D;
#pragma unroll
    for (int q0 = 0; q0 < EPL; q0 += CH) {
      double sv[CH];
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int e = l + 64 * (q0 + u);
        const bool in = (q0 + u < EPL) && (e < DD);
        sv[u] = in ? St[e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int e = l + 64 * (q0 + u);
        if ((q0 + u < EPL) && (e < DD)) {
          const int r = (int)(((unsigned)e * magic) >> 20), cc = e - r * D;
          S.Lm[r * LD + cc] = c * sv[u];
        }
      }
    }
  }
  if (l >= D && pad) S.Lm[l * LD + l] = 1.0;
  // (the vectors go to LDS after the matrix: their stores wait for their loads, which must not hold back the requests for S_t)
  if (pad) { S.mv[l] = v_m; S.bv[l] = v_b; S.am[l] = 0.0; }
  wave_sync();

  VGPA_TRACE_STAMP(0);
  // ---- 1. Cholesky (identical to k_energy_l96)
  bool bad = false;
  double myrd = 1.0;                       // 1 / L[l][l] (padding rows: 1), stored once after the loop
  const double* lrow_b = S.Lm + (4 * b + c4) * LD + r4;
  double* lout_b = S.Lm + (4 * b + r4) * LD + c4;
#pragma unroll
  for (int p = 0; p < NB; p++) {
    const int j0 = 4 * p;
    if (p > 0) {
      const double* brow = S.Lm + (j0 + c4) * LD + r4;
#pragma unroll
      for (int u = 0; p + 4 * u < NB; u++) {
        const bool full = p + 4 * u + 3 < NB;
        const bool rowok = full || (p + 4 * u + b < NB);
        const double* arow = full ? lrow_b + 4 * (p + 4 * u) * LD
                                  : S.Lm + (4 * (rowok ? p + 4 * u + b : NB - 1) + c4) * LD + r4;
        double uacc = 0.0;
#pragma unroll
        for (int kk = 0; kk < p; kk++) uacc = __builtin_amdgcn_mfma_f64_4x4x4f64(arow[4 * kk], brow[4 * kk], uacc, 0, 0, 0);
        if (rowok) lout_b[4 * (p + 4 * u) * LD + j0] -= uacc;
      }
      wave_sync();
    }
    chol_panel_pivots<LD>(S.Lm, myrd, j0, l, li, pad);       // 1/sqrt and sqrt to ~1 ulp, no fp64 divide
    wave_sync();
  }
  if (!CMP && pad) S.rd[l] = myrd;
  wave_sync();
  bad = __any(!(myrd > 0.0 && myrd < __builtin_inf()));
  if (bad) {
    if (l == 0) atomicOr(a.status + prob, 1);
    return;
  }
#pragma unroll
  for (int u = NUE; u < NUU; u++) load_af(u);

  VGPA_TRACE_STAMP(1);
  // ---- 2. A.m and G = A.L on the matrix cores, results stay in the accumulators (row 16u + 4b + r4, column 4J + c4)
  double amr[NUU], gacc[NUU][NB];
#pragma unroll
  for (int u = 0; u < NUU; u++) {
    amr[u] = 0.0;
#pragma unroll
    for (int J = 0; J < NB; J++) gacc[u][J] = 0.0;
  }
#pragma unroll
  for (int pass = 0; pass < (NUE < NUU ? 2 : 1); pass++) {    // units 0 .. NUE-1, then the late unit (its own reads of L)
    const int u0 = pass == 0 ? 0 : NUE, u1 = pass == 0 ? NUE : NUU;
#pragma unroll
    for (int K = 0; K < NB; K++) {
      const double mk = S.mv[4 * K + r4];                       // B-operand [k = r4][j]: m in every column
#pragma unroll
      for (int u = u0; u < u1; u++) amr[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[u][K], mk, amr[u], 0, 0, 0);
      const double* lk = S.Lm + (4 * K + r4) * LD + c4;          // B-operand [k = r4][j = c4] = L[4K + r4][4J + c4], K >= J
#pragma unroll
      for (int J = 0; J <= K; J++) {
        const double lf = lk[4 * J];
#pragma unroll
        for (int u = u0; u < u1; u++) gacc[u][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[u][K], lf, gacc[u][J], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NUU; u++) {
    const int i = 16 * u + 4 * b + r4;
    if (c4 == 0 && i < D) { S.am[i] = amr[u]; if (a.Am) a.Am[o * D + i] = amr[u]; }
  }

  VGPA_TRACE_STAMP(2);
  // ---- 3. residuals of the sigma points m +- L[:, j] in accumulator layout, rows 2 .. D-2
  // park rows 0, 1, D-1 of G for the boundary pass first (only the units that can hold them: compile-time test), so
  // that the residual loop below is ONE basic block (with branches in between, the compiler sinks all the arithmetic
  // below the last branch and keeps 8 NB NUU registers of LDS reads alive)
#pragma unroll
  for (int u = 0; u < NUU; u++) {
    if (u == 0 || (16 * u <= Dp - 1 && 16 * u + 15 >= Dp - 4)) {
      const int i = 16 * u + 4 * b + r4;
      if constexpr (CMP) {
        if (i == D - 1) {
#pragma unroll
          for (int J = 0; J < NB; J++) *cg2(4 * J + c4) = gacc[u][J];
        }
      }
      double* gdst = (i == 0) ? S.dl : ((i == 1) ? S.qq : ((i == D - 1 && !CMP) ? gb2 : nullptr));
      if (gdst) {
#pragma unroll
        for (int J = 0; J < NB; J++) gdst[4 * J + c4] = gacc[u][J];
      }
    }
  }
  // (round 3) The residuals of the sigma-point PAIR m +- L[:, j] are evaluated as E +- O: with dm = m_{i+1} - m_{i-2},
  // dl = l_{i+1} - l_{i-2} (l = L[:, j]) the drift's product (dm +- dl)(m_{i-1} +- l_{i-1}) splits into an even part
  // dm m_{i-1} + dl l_{i-1} and an odd part dm l_{i-1} + dl m_{i-1}, so
  //     r_+- = E +- O,   E = [dm m_{i-1} - m_i + theta + (A m)_i - b_i] + dl l_{i-1},   O = dm l_{i-1} + dl m_{i-1} - l_i + G_ij,
  // and the two weighted sums the kernel needs are  sum s_i r_+-^2 = S1 +- 2 S2  with S1 = sum s_i (E^2 + O^2), S2 = sum s_i E O:
  // 10 fused operations per element pair instead of 26 (the kernel is bound by the instructions it issues: ~30 k SIMD cycles per
  // grid point, 6 k of them fp64 vector arithmetic), and v_+ - v_- no longer comes from the difference of two large sums.  The
  // contraction a*b+c -> fma is allowed in this block (tolerance 1e-6 asked, 1e-9 tested; the build's default is off).
  double vp[NB], vm[NB];                                          // S1, S2 of the interior rows
#pragma unroll
  for (int J = 0; J < NB; J++) { vp[J] = 0.0; vm[J] = 0.0; }
  {
#pragma clang fp contract(fast)
#pragma unroll
  for (int u = 0; u < NUU; u++) {
    const int i = 16 * u + 4 * b + r4;
    const bool ok = (i >= 2) && (i <= D - 2);
    const int ir = ok ? i : 2;                                  // masked lanes read a valid interior row, weight 0
    const double mm2 = S.mv[ir - 2], mm1 = S.mv[ir - 1], m0 = S.mv[ir], m1 = S.mv[ir + 1];
    const double sgr = __shfl(v_sg, ir, 64);                    // Sigma^-1[ir][ir] lives in lane ir's register
    const double bvi = S.bv[ir], sgi = ok ? sgr : 0.0, ami = amr[u];
    const double dm = m1 - mm2;
    const double base = dm * mm1 - m0 + theta + ami - bvi;
    const double* lp = S.Lm + (ir - 2) * LD + c4;
#pragma unroll
    for (int J = 0; J < NB; J++) {
      const double e2 = lp[4 * J], e1 = lp[LD + 4 * J], e0 = lp[2 * LD + 4 * J], ep = lp[3 * LD + 4 * J];
      const double gi = gacc[u][J];
      const double dl = ep - e2;
      const double ev = base + dl * e1;
      const double od = dm * e1 + (dl * mm1 + (gi - e0));
      vp[J] = vp[J] + sgi * (ev * ev + od * od);
      vm[J] = vm[J] + sgi * (ev * od);
      if (J == NB / 2 - 1 || J == NB - 1) __builtin_amdgcn_sched_barrier(0);    // keep the scheduler from hoisting all 4 NB LDS reads (registers)
    }
  }
  }
  // sum over the rows: ones-MFMA adds the four r4 of a block slot, the four block slots b go through LDS
  {
    double rp[NB], rm[NB];
#pragma unroll
    for (int J = 0; J < NB; J++) {
      rp[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, vp[J], 0.0, 0, 0, 0);
      rm[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, vm[J], 0.0, 0, 0, 0);
    }
    if constexpr (CMP) {
      // the four block slots of a row of 16 lanes are summed by rotations (every lane ends with the total of its column), the lanes
      // of block slot 0 of the first row write the two rows of totals -- behind a synchronisation: the residual loop of some lanes
      // may still be reading the zeros the totals replace
#pragma unroll
      for (int J = 0; J < NB; J++) {       // (two rotations: slot b + slot b-1, then that pair + the pair two slots away)
        const double p1 = rp[J] + row_ror_blocks<1>(rp[J]), m1 = rm[J] + row_ror_blocks<1>(rm[J]);
        rp[J] = p1 + row_ror_blocks<2>(p1);
        rm[J] = m1 + row_ror_blocks<2>(m1);
      }
      wave_sync();
      if (r4 == 0 && b == 0) {
#pragma unroll
        for (int J = 0; J < NB; J++) {
          *cps(0, 4 * J + c4) = rp[J];
          *cps(1, 4 * J + c4) = rm[J];
        }
      }
    } else
    if (r4 == 0) {
#pragma unroll
      for (int J = 0; J < NB; J++) {
        pv[b * Dp + 4 * J + c4] = rp[J];
        pw[b * Dp + 4 * J + c4] = rm[J];
      }
    }
  }
  wave_sync();
  VGPA_TRACE_STAMP(3);

  double vplus = 0.0, vminus = 0.0, v0 = 0.0;
  {
    auto col_of = [&](int q) { return q == 0 ? 0 : (q <= D ? q - 1 : q - 1 - D); };
    auto sgn_of = [&](int q) { return q == 0 ? 0.0 : (q <= D ? 1.0 : -1.0); };
    auto chi = [&](int q, int i) { return S.mv[i] + sgn_of(q) * S.Lm[i * LD + col_of(q)]; };
    const int jc = act ? l : 0;
    {
      double s1, s2;
      if constexpr (CMP) { s1 = *cps(0, jc); s2 = *cps(1, jc); }
      else {
        s1 = (pv[jc] + pv[Dp + jc]) + (pv[2 * Dp + jc] + pv[3 * Dp + jc]);
        s2 = (pw[jc] + pw[Dp + jc]) + (pw[2 * Dp + jc] + pw[3 * Dp + jc]);
      }
      vplus = s1 + 2.0 * s2;                       // sum s_i r_+^2 over the interior rows
      vminus = s1 - 2.0 * s2;
    }
    // rows 0, 1 and D-1: the flat np.roll of the reference (quirk Q1) takes their neighbours from the sigma points
    // p-1 and p+1.  Window over the flat index w = p*D + i: X(w-2), X(w-1), X(w), X(w+1).
    const int pP = 1 + jc, pM = 1 + D + jc;
    const int pPm = pP - 1, pPp = pP + 1;
    const int pMm = pM - 1, pMp = wrap(pM + 1, M);
    {
      const double P0 = chi(pPm, D - 2), P1 = chi(pPm, D - 1), P2 = chi(pP, 0), P3 = chi(pP, 1), P4 = chi(pP, 2);
      const double N0 = chi(pMm, D - 2), N1 = chi(pMm, D - 1), N2 = chi(pM, 0), N3 = chi(pM, 1), N4 = chi(pM, 2);
      const double Q0 = chi(pP, D - 3), Q1 = chi(pP, D - 2), Q2 = chi(pP, D - 1), Q3 = chi(pPp, 0);
      const double R0 = chi(pM, D - 3), R1 = chi(pM, D - 2), R2 = chi(pM, D - 1), R3 = chi(pMp, 0);
      const double g0 = S.dl[jc], g1 = S.qq[jc], g2 = CMP ? *cg2(jc) : gb2[jc];
      {
        const double ami = S.am[0], bvi = S.bv[0], sgi = lane_value(v_sg, 0);
        const double ra = ((P3 - P0) * P1 - P2 + theta) + (ami + g0) - bvi;
        const double rb = ((N3 - N0) * N1 - N2 + theta) + (ami - g0) - bvi;
        vplus = __builtin_fma(sgi, ra * ra, vplus);
        vminus = __builtin_fma(sgi, rb * rb, vminus);
      }
      {
        const double ami = S.am[1], bvi = S.bv[1], sgi = lane_value(v_sg, 1);
        const double ra = ((P4 - P1) * P2 - P3 + theta) + (ami + g1) - bvi;
        const double rb = ((N4 - N1) * N2 - N3 + theta) + (ami - g1) - bvi;
        vplus = __builtin_fma(sgi, ra * ra, vplus);
        vminus = __builtin_fma(sgi, rb * rb, vminus);
      }
      {
        const double ami = S.am[D - 1], bvi = S.bv[D - 1], sgi = lane_value(v_sg, D - 1);
        const double ra = ((Q3 - Q0) * Q1 - Q2 + theta) + (ami + g2) - bvi;
        const double rb = ((R3 - R0) * R1 - R2 + theta) + (ami - g2) - bvi;
        vplus = __builtin_fma(sgi, ra * ra, vplus);
        vminus = __builtin_fma(sgi, rb * rb, vminus);
      }
    }
    // the mean point: lane i holds term i of the sum
    if (act) {
      const int i = l;
      const double xm2 = (i >= 2) ? S.mv[i - 2] : chi(M - 1, D - 2 + i);
      const double xm1 = (i >= 1) ? S.mv[i - 1] : chi(M - 1, D - 1);
      const double x1 = (i + 1 < D) ? S.mv[i + 1] : chi(1, 0);
      const double r0 = ((x1 - xm2) * xm1 - S.mv[i] + theta) + S.am[i] - S.bv[i];
      v0 = v_sg * (r0 * r0);
    }
    v0 = wave_sum(v0);
  }
  const double w0 = kappa / c, w1 = 1.0 / (2.0 * c);
  const double e_part = act ? (vplus + vminus) : 0.0;
  const double e_t = 0.5 * (w0 * v0 + w1 * wave_sum(e_part));
  wave_sync();                                   // every lane has read the partial sums: xdiag may overwrite them
  if (pad) {                                     // (padding entries back to zero: the parked rows of G were here)
    S.dl[l] = act ? w1 * (vplus - vminus) : 0.0;
    S.qq[l] = act ? 0.5 * c * (w1 * (vplus + vminus)) - e_t : 0.0;
  }
  if (l == 0) a.e_t[o] = e_t;
  // inverses of the 4x4 diagonal blocks of L (lane I < NB)
  const int l4 = l < NB ? 4 * l : 0;
  const double rd0 = CMP ? __shfl(myrd, l4, 64) : 0.0, rd1 = CMP ? __shfl(myrd, l4 + 1, 64) : 0.0;
  const double rd2 = CMP ? __shfl(myrd, l4 + 2, 64) : 0.0, rd3 = CMP ? __shfl(myrd, l4 + 3, 64) : 0.0;
  if (l < NB) {
    const double* tb = S.Lm + (4 * l) * LD + 4 * l;
    const double x00 = CMP ? rd0 : S.rd[4 * l], x11 = CMP ? rd1 : S.rd[4 * l + 1], x22 = CMP ? rd2 : S.rd[4 * l + 2], x33 = CMP ? rd3 : S.rd[4 * l + 3];
    const double t10 = tb[LD], t20 = tb[2 * LD], t21 = tb[2 * LD + 1], t30 = tb[3 * LD], t31 = tb[3 * LD + 1], t32 = tb[3 * LD + 2];
    const double x10 = -(t10 * x00) * x11;
    const double x21 = -(t21 * x11) * x22;
    const double x32 = -(t32 * x22) * x33;
    const double x20 = -(t20 * x00 + t21 * x10) * x22;
    const double x31 = -(t31 * x11 + t32 * x21) * x33;
    const double x30 = -(t30 * x00 + t31 * x10 + t32 * x20) * x33;
    if constexpr (CMP) {
      double* x0 = cxd(l, 0, 0);
      x0[0] = x00; x0[1] = 0.0; x0[2] = 0.0; x0[3] = 0.0;
      x0[LD] = x10; x0[LD + 1] = x11; x0[LD + 2] = 0.0; x0[LD + 3] = 0.0;
      x0[2 * LD] = x20; x0[2 * LD + 1] = x21; x0[2 * LD + 2] = x22; x0[2 * LD + 3] = 0.0;
      x0[3 * LD] = x30; x0[3 * LD + 1] = x31; x0[3 * LD + 2] = x32; x0[3 * LD + 3] = x33;
    } else {
    double* xo = xdiag + 16 * l;
    xo[0] = x00; xo[1] = 0.0; xo[2] = 0.0; xo[3] = 0.0;
    xo[4] = x10; xo[5] = x11; xo[6] = 0.0; xo[7] = 0.0;
    xo[8] = x20; xo[9] = x21; xo[10] = x22; xo[11] = 0.0;
    xo[12] = x30; xo[13] = x31; xo[14] = x32; xo[15] = x33;
    }
  }
  wave_sync();

  VGPA_TRACE_STAMP(4);
#ifndef VGPA_ENERGY_PF
#define VGPA_ENERGY_PF 4
#endif
  if (VGPA_ENERGY_PF == 4 && TPW > 1 && spk && t + 1 < t_last) request(t + 1);      // the next grid point's operands travel under phases 4 and 5
  // ---- 4. X = L^-1 IN PLACE by blocked forward substitution on the matrix cores (see k_energy_l96): block-row I of X
  //         overwrites block-row I of L, which only step I reads; LDS operations of one wave execute in order.
  const double* l4_a = S.Lm + c4 * LD + r4;
  const double* xd_a = xdiag + 4 * c4 + r4;
  const double* xd_d = xdiag + 4 * r4 + c4;
#pragma unroll
  for (int I = 0; I < NB; I++) {
    const double* arow = l4_a + 4 * I * LD;
    const double xd = CMP ? *cxd(I, c4, r4) : xd_a[16 * I];
    const double xdd = CMP ? *cxd(I, r4, c4) : xd_d[16 * I];
    double xo[NUU];
    // all products of the block-row first (they read L[I][:]), then the stores (they overwrite it)
#pragma unroll
    for (int u = 0; u < NUU; u++) {
      const int Jb = 4 * u + b;
      const bool colok = (4 * u + 3 < NB) || (Jb < NB);
      const double* xcol = S.Lm + r4 * LD + 4 * (colok ? Jb : NB - 1) + c4;
      xo[u] = 0.0;
      if (4 * u <= I) {
        double tacc = 0.0;
#pragma unroll
        for (int K = 4 * u; K < I; K++) tacc = __builtin_amdgcn_mfma_f64_4x4x4f64(arow[4 * K], xcol[4 * K * LD], tacc, 0, 0, 0);
        const double prod = __builtin_amdgcn_mfma_f64_4x4x4f64(xd, tacc, 0.0, 0, 0, 0);
        xo[u] = (Jb < I) ? -prod : ((Jb == I) ? xdd : 0.0);
      }
    }
#pragma unroll
    for (int u = 0; u < NUU; u++) {
      const int Jb = 4 * u + b;
      const bool colok = (4 * u + 3 < NB) || (Jb < NB);
      // (compact layout: the blocks right of the diagonal are exact zeros already -- and the far ones hold the diagonal blocks' inverses)
      if (colok && (!CMP || Jb <= I)) S.Lm[(4 * I + r4) * LD + 4 * Jb + c4] = xo[u];
    }
    wave_sync();
  }

  VGPA_TRACE_STAMP(5);
  if (VGPA_ENERGY_PF == 5 && TPW > 1 && spk && t + 1 < t_last) request(t + 1);
  // ---- 5. dE/dm = (c/2) X^T delta ; dE/dS = (c/2) X^T diag(q) X
  {
    double s = 0.0;
    if constexpr (CMP) {               // (rows 0 .. 7 of the columns >= 16 hold the diagonal blocks' inverses, not the zeros of X: masked)
      const bool z0 = li < 4, z1 = li < 8;
#pragma unroll
      for (int k = 0; k < 8; k++) s = __builtin_fma((k < 4 ? z0 : z1) ? S.Lm[k * LD + li] : 0.0, S.dl[k], s);
      for (int k = 8; k < Dp; k++) s = __builtin_fma(S.Lm[k * LD + li], S.dl[k], s);
    } else
    for (int k = 0; k < Dp; k++) s = __builtin_fma(S.Lm[k * LD + li], S.dl[k], s);
    if (act) a.dEm[o * D + l] = 0.5 * c * s;
  }
  double* ds = a.dEs + o * D * D;
  {
    double acc[WaveGemmGeo<NB>::NU];
    if (a.ds_packed) {                // the lower triangle, packed (a row's 16-column segment of a unit is contiguous): the units above the
      using gg = WaveGemmGeo<NB>;     // diagonal are not computed
      double* dsp = a.dEs + o * PK;
      wave_gemm<NB, 1, 2>(S.Lm, S.Lm, LD, acc, S.qq);
#pragma unroll
      for (int u = 0; u < gg::NU; u++) {
        if (u < NB * gg::NQ && 4 * (u / NB) > u % NB) continue;
        if (u >= NB * gg::NQ && (u - NB * gg::NQ) * gg::G + gg::G - 1 < 4 * gg::NQ) continue;
        int row, col; bool ok;
        wave_gemm_elem<NB>(u, row, col, ok);
        if (ok && row < D && col <= row) dsp[tri_off(row) + col] = 0.5 * c * acc[u];
      }
    } else if (a.ds_upper) {          // (units below the diagonal are not even computed: -22 % of the SYRK's products at NB = 10)
      wave_gemm<NB, 1, 1>(S.Lm, S.Lm, LD, acc, S.qq);
#pragma unroll
      for (int u = 0; u < WaveGemmGeo<NB>::NU; u++) {
        if (u < NB * WaveGemmGeo<NB>::NQ && 4 * (u / NB) + 3 < u % NB) continue;
        int row, col; bool ok;
        wave_gemm_elem<NB>(u, row, col, ok);
        if (ok && row < D && col < D && row <= col) ds[row * D + col] = 0.5 * c * acc[u];
      }
    } else {
      wave_gemm<NB, 1>(S.Lm, S.Lm, LD, acc, S.qq);
#pragma unroll
      for (int u = 0; u < WaveGemmGeo<NB>::NU; u++) {
        int row, col; bool ok;
        wave_gemm_elem<NB>(u, row, col, ok);
        if (ok && row < D && col < D) ds[row * D + col] = 0.5 * c * acc[u];
      }
    }
  }

  VGPA_TRACE_STAMP(6);
  // ---- <f> and optionally dense <df/dx>
  if (act) {
    const int i = l, ip1 = wrap(i + 1, D), im1 = wrap(i - 1, D), im2 = wrap(i - 2, D);
    const double cxx = sxa - sxb;
    a.Ef[o * D + i] = cxx + (S.mv[ip1] - S.mv[im2]) * S.mv[im1] - S.mv[i] + theta;
  }
  if (a.Edf) {
    double* ed = a.Edf + o * D * D;
    for (int e = l; e < D * D; e += 64) {
      const int k = e / D, j = e - k * D;
      const int kp1 = wrap(k + 1, D), km1 = wrap(k - 1, D), km2 = wrap(k - 2, D);
      double v = 0.0;
      if (j == k) v = -1.0;
      if (j == kp1) v = S.mv[km1];
      if (j == km2) v = -S.mv[km1];
      if (j == km1) v = S.mv[kp1] - S.mv[km2];
      ed[e] = v;
    }
  }
  VGPA_TRACE_STAMP(7);
  if (TPW == 1) break;
  if (VGPA_ENERGY_PF == 7 && spk && t + 1 < t_last) request(t + 1);
  t++;
  if (t >= t_last) break;
  o++;
  At += (size_t)D * D;
  St += spk ? PK : D * D;
  wave_sync();                                   // every lane is done with this grid point's LDS before the next one is staged
  }
}

// dense <df/dx> only (used by vgpa_fetch(EDF) when the fused sweep skipped it)
__global__ void __launch_bounds__(NT) k_edf_dense(EnergyArgs a) {
  const int D = a.D;
  const int t = blockIdx.x, prob = blockIdx.y;
  const size_t o = (size_t)prob * a.Np + t;
  const double* mv = a.m + o * D;
  double* ed = a.Edf + o * D * D;
  for (int e = threadIdx.x; e < D * D; e += NT) {
    const int k = e / D, j = e - k * D;
    double v = 0.0;
    if (a.model == VGPA_MODEL_L96) {
      const int kp1 = wrap(k + 1, D), km1 = wrap(k - 1, D), km2 = wrap(k - 2, D);
      if (j == k) v = -1.0;
      if (j == kp1) v = mv[km1];
      if (j == km2) v = -mv[km1];
      if (j == km1) v = mv[kp1] - mv[km2];
    } else if (a.model == VGPA_MODEL_L63) {
      const double vS = a.theta[0], vR = a.theta[1], vB = a.theta[2];
      const double tab[9] = {-vS, vS, 0.0, vR - mv[2], -1.0, -mv[0], mv[1], mv[0], -vB};
      v = tab[e];
    } else if (a.model == VGPA_MODEL_OU) {
      v = -a.theta[0];
    } else {
      v = 4.0 * (a.theta[0] - 3.0 * (mv[0] * mv[0] + a.S[o]));
    }
    ed[e] = v;
  }
}

}  // namespace
}  // namespace vgpa

namespace vgpa {

hipError_t launch_energy(const EnergyArgs& a, hipStream_t st) {
  if (a.model == VGPA_MODEL_OU || a.model == VGPA_MODEL_DW) {
    dim3 grid((a.Np + NT - 1) / NT, a.batch);
    hipLaunchKernelGGL(k_energy_1d, grid, dim3(NT), 0, st, a);
  } else if (a.model == VGPA_MODEL_L63) {
    dim3 grid((a.Np + 63) / 64, a.batch);
    hipLaunchKernelGGL(k_energy_l63, grid, dim3(64), 0, st, a);
  } else if (a.model == VGPA_MODEL_L96) {
    if (a.D < 4 || a.D > kMaxSmallD) return hipErrorInvalidValue;
    const bool one_matrix = (a.hyp == nullptr);      // the hyper-parameter integrands need the second LDS matrix
    size_t lds = (one_matrix ? l96r_lds_doubles(a.D) : l96_lds_doubles(a.D)) * sizeof(double);
    const long long nwaves = (long long)a.Np * a.batch;
    if (nwaves > 0x7fffffffLL) return hipErrorInvalidValue;
#if defined(VGPA_EXPERIMENTS) && VGPA_ENERGY_TPW > 1     /* EXPERIMENT: packed S_t, persistent waves, see k_energy_l96_r */
#define VGPA_L96_TPW(NBV)                                                                                           \
    if constexpr (NBV == 9 || NBV == 10) {                                                                          \
      if (one_matrix && a.s_packed) {                                                                               \
        constexpr int TPW = kEnergyTPW;                                                                             \
        (void)hipFuncSetAttribute((const void*)k_energy_l96_r<NBV, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        const long long nw = (long long)((a.Np + TPW - 1) / TPW) * a.batch;                                          \
        hipLaunchKernelGGL((k_energy_l96_r<NBV, TPW>), dim3((unsigned)nw), dim3(64), lds, st, a);                    \
        break;                                                                                                      \
      }                                                                                                             \
    }
#else
#define VGPA_L96_TPW(NBV)
#endif
#define VGPA_L96_CASE(NBV)                                                                                          \
  case NBV:                                                                                                         \
    if (lds > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute(one_matrix ? (const void*)k_energy_l96_r<NBV> : (const void*)k_energy_l96<NBV>,       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
    VGPA_L96_TPW(NBV)                                                                                               \
    if (one_matrix) hipLaunchKernelGGL(k_energy_l96_r<NBV>, dim3((unsigned)nwaves), dim3(64), lds, st, a);          \
    else hipLaunchKernelGGL(k_energy_l96<NBV>, dim3((unsigned)nwaves), dim3(64), lds, st, a);                        \
    break;
    switch ((a.D + 3) / 4) {
      VGPA_L96_CASE(1) VGPA_L96_CASE(2) VGPA_L96_CASE(3) VGPA_L96_CASE(4) VGPA_L96_CASE(5) VGPA_L96_CASE(6)
      VGPA_L96_CASE(7) VGPA_L96_CASE(8) VGPA_L96_CASE(9) VGPA_L96_CASE(10) VGPA_L96_CASE(11) VGPA_L96_CASE(12)
      VGPA_L96_CASE(13) VGPA_L96_CASE(14) VGPA_L96_CASE(15) VGPA_L96_CASE(16)
      default: return hipErrorInvalidValue;
    }
#undef VGPA_L96_CASE
#undef VGPA_L96_TPW
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_edf(const EnergyArgs& a, hipStream_t st) {
  dim3 grid(a.Np, a.batch);
  hipLaunchKernelGGL(k_edf_dense, grid, dim3(NT), 0, st, a);
  return hipGetLastError();
}

}  // namespace vgpa

#ifdef VGPA_ENERGY_TRACE
extern "C" int vgpa_debug_energy_trace(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vgpa::g_energy_trace), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(vgpa::g_energy_trace), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
