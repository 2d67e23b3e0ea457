// Time stepping for D <= 4 (Ornstein-Uhlenbeck, double well: D = 1; Lorenz-63: D = 3): ONE LANE per problem, the
// whole state in registers, no LDS, no barriers.  A batch of independent problems fills the lanes; the recursion of a
// problem is sequential in t, so its latency is one lane's instruction latency (comparable to the workgroup-per-problem
// kernel of ode_generic.hip at D = 3), but 64 problems share a wave instead of each occupying a workgroup of four
// waves that synchronises six times per step: BASELINE configs[1] (Lorenz-63, RK4, Np = 1001), 65536 problems:
// forward + backward 236 ms -> see DESIGN.md s.4.3.
//
// Same arithmetic, in the same order, as ode_generic.hip (which restates the reference):
//   forward : euler.py:27-92, heun.py:28-111, runge_kutta2.py:25-102 (incl. quirk Q2 at :96), runge_kutta4.py:25-113
//   backward: euler.py:94-154, heun.py:113-190, runge_kutta2.py:104-194, runge_kutta4.py:115-211
//             (jumps added after the step, Q9; f_lam uses A.lam, Q3)
//   RHS     : f_S = -A S - S A^T + Sigma (ode_solver.py:60), f_Psi = -G + Psi A + A^T Psi (ode_solver.py:94)
// A_{t+-1}, b / dE of the next step are requested one step ahead.
//
// Round 4 -- memory path.  A lane's operands sit len_x doubles away from its neighbour's, so a per-lane load is 64 cache-line
// requests per instruction and the kernels were bound by the request rate of the texture path (0.15 of the HBM roofline at
// 65536 Lorenz-63 problems), not by bytes.  k_fwd_lane / k_sweep_lane move every time-indexed stream through LDS in CHUNKS of T
// grid points: the wave reads the chunk of its 64 problems with flat, coalesced loads (lane i takes double i of the
// concatenated per-problem pieces: 5-10 line requests per instruction), parks it in registers for the T steps of the previous
// chunk, drops it into LDS rows (row p = problem p), and each lane consumes its own row; results take the consumed slots and
// leave the same way.  k_sweep_lane is the FUSED backward pass of the models with closed-form moments (OU, double well,
// Lorenz-63: BASELINE configs[0] / [1]): a lane re-evaluates the energy terms of grid point t-1 from (A, b, m, S) in registers
// (energy_small.h), steps (lam, Psi), assembles gLa / gLb of t-1 and carries the trapezoid of E_sde -- dEsde_dm, dEsde_dS,
// <f>, E_sde(t), lam_t, Psi_t never touch HBM (vgpa_fetch materialises them on demand through the separate kernels).
#include <cstdlib>

#include "vgpa_internal.h"
#include "energy_small.h"

namespace vgpa {
namespace {

constexpr int NTS = 64;   // lanes (problems) per workgroup

template <int D>
__device__ __forceinline__ void ld_mat(const double* __restrict__ p, double (&v)[D * D]) {
#pragma unroll
  for (int e = 0; e < D * D; e++) v[e] = p[e];
}
template <int D>
__device__ __forceinline__ void ld_vec(const double* __restrict__ p, double (&v)[D]) {
#pragma unroll
  for (int e = 0; e < D; e++) v[e] = p[e];
}
template <int D>
__device__ __forceinline__ void st_mat(double* __restrict__ p, const double (&v)[D * D]) {
#pragma unroll
  for (int e = 0; e < D * D; e++) p[e] = v[e];
}
template <int D>
__device__ __forceinline__ void st_vec(double* __restrict__ p, const double (&v)[D]) {
#pragma unroll
  for (int e = 0; e < D; e++) p[e] = v[e];
}

// r = f_S(X; AB) = -AB.X - X.AB^T + Sigma
template <int D>
__device__ __forceinline__ void rhs_fwd(const double (&AB)[D * D], const double (&X)[D * D], const double (&sig)[D * D],
                                        double (&r)[D * D]) {
#pragma unroll
  for (int i = 0; i < D; i++)
#pragma unroll
    for (int j = 0; j < D; j++) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < D; k++) {
        s1 = __builtin_fma(AB[i * D + k], X[k * D + j], s1);
        s2 = __builtin_fma(X[i * D + k], AB[j * D + k], s2);
      }
      r[i * D + j] = (-s1 - s2) + sig[i * D + j];
    }
}

// r = f_Psi(G, AB, X) = -G + X.AB + AB^T.X
template <int D>
__device__ __forceinline__ void rhs_bwd(const double (&AB)[D * D], const double (&X)[D * D], const double (&g)[D * D],
                                        double (&r)[D * D]) {
#pragma clang fp contract(fast)
#pragma unroll
  for (int i = 0; i < D; i++)
#pragma unroll
    for (int j = 0; j < D; j++) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < D; k++) {
        s1 = __builtin_fma(X[i * D + k], AB[k * D + j], s1);
        s2 = __builtin_fma(AB[k * D + i], X[k * D + j], s2);
      }
      r[i * D + j] = (-g[i * D + j] + s1) + s2;
    }
}

template <int D>
__device__ __forceinline__ void matvec(const double (&AB)[D * D], const double (&x)[D], double (&y)[D]) {
#pragma unroll
  for (int i = 0; i < D; i++) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; k++) s = __builtin_fma(AB[i * D + k], x[k], s);
    y[i] = s;
  }
}

template <int D>
__device__ __forceinline__ void mid_mat(const double (&a0)[D * D], const double (&a1)[D * D], double (&m)[D * D]) {
#pragma unroll
  for (int e = 0; e < D * D; e++) m[e] = 0.5 * (a0[e] + a1[e]);
}

// ---- coalesced chunks: N doubles per problem and chunk, 64 problems per wave -> N doubles per lane ----------------------------
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N>
constexpr int row_stride() { return N | 1; }      // odd: a lane's row reads are spread over the banks

// How the 64 lanes of an instruction cover the chunk: G consecutive doubles (G divides N) of each of 64 / G problems (rounded UP:
// the last lanes start the next group's first problem -- the same data the next group's instruction moves again, a benign
// duplicate); a problem's N doubles take R = N / G instructions, the 64 problems NG groups.  Affine in the lane -- lane l always
// works on (problem l / G, double l % G) of its instruction's block -- so that addresses are one per-lane offset plus uniform
// terms (not one precomputed address pair per instruction, which is what the register allocator makes of a flat index), and no
// access needs a predicate: out-of-range problems / grid points are CLAMPED to valid ones, whose data they duplicate.
template <int N>
struct ChunkMap {
  static constexpr int cost(int g) { return (N % g) ? (1 << 20) : ((NTS + NTS / g - 1) / (NTS / g)) * (N / g); }
  static constexpr int pick() {
    int best = 1;
    for (int g = 2; g <= 16; g++)
      if (cost(g) <= cost(best)) best = g;
    return best;
  }
  static constexpr int G = pick();
  static constexpr int R = N / G;
  static constexpr int PG = NTS / G;                       // whole problems per instruction
  static constexpr int NG = (NTS + PG - 1) / PG;
  static constexpr int NI = NG * R;                        // instructions per chunk = doubles a lane holds between request and LDS
  static constexpr int LDS = NTS * (N | 1);                // (lanes that fall behind problem 63 work on problem 63 again)
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Requests the chunk [first, first + N) of every problem of the wave.  base_u: the stream's first problem of this wave
// (wave-uniform); problem p at base_u + p * pstride, `limit` doubles each.  Indices outside [0, limit) are clamped (their values
// are never used), problems beyond the batch repeat the last valid one.  32-bit offsets: 64 * pstride * 8 < 2^32 (launcher).
template <int N>
__device__ __forceinline__ void chunk_request(const double* __restrict__ base_u, unsigned pstride, int nvalid, long first, long limit,
                                              double (&v)[ChunkMap<N>::NI]) {
  using M = ChunkMap<N>;
  const unsigned lane = threadIdx.x;
  const int pl = (int)(lane / M::G), el = (int)(lane % M::G);
  const char* bu = reinterpret_cast<const char*>(base_u + first);      // wave-uniform; only ever dereferenced at clamped offsets
  const int lo = (int)(-first), hi = (int)(limit - 1 - first);
#pragma unroll
  for (int jg = 0; jg < M::NG; jg++)
#pragma unroll
    for (int r = 0; r < M::R; r++) {
      int p = jg * M::PG + pl;
      p = p < nvalid ? p : nvalid - 1;
      const int e = clampi(r * M::G + el, lo, hi);
      v[jg * M::R + r] = *reinterpret_cast<const double*>(bu + ((unsigned)p * pstride + (unsigned)e) * 8u);
    }
}

template <int N>
__device__ __forceinline__ void chunk_to_lds(double* __restrict__ lds, const double (&v)[ChunkMap<N>::NI]) {
  using M = ChunkMap<N>;
  const unsigned lane = threadIdx.x;
  const int pl = (int)(lane / M::G), el = (int)(lane % M::G);
#pragma unroll
  for (int jg = 0; jg < M::NG; jg++) {
    const int p = (jg * M::PG + NTS / M::G > NTS - 1) ? (jg * M::PG + pl < NTS ? jg * M::PG + pl : NTS - 1) : jg * M::PG + pl;
    double* mine = lds + p * row_stride<N>() + el;
#pragma unroll
    for (int r = 0; r < M::R; r++) mine[r * M::G] = v[jg * M::R + r];
  }
}

// results leave the way the operands came; lanes whose (problem, grid point) is out of range store a valid neighbour's value to
// that neighbour's address a second time
template <int N>
__device__ __forceinline__ void chunk_flush(const double* __restrict__ lds, double* __restrict__ base_u, unsigned pstride, int nvalid,
                                            long first, long limit) {
  using M = ChunkMap<N>;
  const unsigned lane = threadIdx.x;
  const int pl = (int)(lane / M::G), el = (int)(lane % M::G);
  char* bu = reinterpret_cast<char*>(base_u + first);
  const int lo = (int)(-first), hi = (int)(limit - 1 - first);
#pragma unroll
  for (int jg = 0; jg < M::NG; jg++)
#pragma unroll
    for (int r = 0; r < M::R; r++) {
      int p = jg * M::PG + pl;
      p = p < nvalid ? p : nvalid - 1;
      const int e = clampi(r * M::G + el, lo, hi);
      const double val = lds[p * row_stride<N>() + e];
      *reinterpret_cast<double*>(bu + ((unsigned)p * pstride + (unsigned)e) * 8u) = val;
    }
}

// x / 6 without the division sequence: q = x * RN(1/6), one exact residual, one correction (Markstein): the correctly rounded
// quotient for every x in the normal range (RK4's dt (k1 + 2 k2 + 2 k3 + k4) / 6.0: runge_kutta4.py:107-108, 205-206)
__device__ __forceinline__ double div6(double x) {
  const double y = 1.0 / 6.0;
  const double q = x * y;
  const double r = __builtin_fma(-6.0, q, x);
  return __builtin_fma(r, y, q);
}

// ------------------------------------------------------------------------------------------------
// one forward step k -> k+1: (sk, mk) in place; operands of the start point (A0, b0) and of the end point (A1, b1)
template <int METHOD, int D>
__device__ __forceinline__ void fwd_step(const double (&A0)[D * D], const double (&A1)[D * D], const double (&b0)[D], const double (&b1)[D],
                                         const double (&sig)[D * D], double dt, double (&sk)[D * D], double (&mk)[D]) {
  constexpr int DD = D * D;
  const double h = 0.5 * dt;
  double r[DD], y[D], X[DD], xv[D];
  if (METHOD == VGPA_ODE_EULER) {
    rhs_fwd<D>(A0, sk, sig, r);
    matvec<D>(A0, mk, y);
#pragma unroll
    for (int e = 0; e < DD; e++) sk[e] = sk[e] + r[e] * dt;
#pragma unroll
    for (int i = 0; i < D; i++) mk[i] = mk[i] + (-y[i] + b0[i]) * dt;
  } else if (METHOD == VGPA_ODE_HEUN) {
    double acc1[DD], pm[D];
    rhs_fwd<D>(A0, sk, sig, r);
    matvec<D>(A0, mk, y);
#pragma unroll
    for (int i = 0; i < D; i++) { pm[i] = -y[i] + b0[i]; xv[i] = mk[i] + pm[i] * dt; }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc1[e] = r[e]; X[e] = sk[e] + r[e] * dt; }
    rhs_fwd<D>(A1, X, sig, r);
    matvec<D>(A1, xv, y);
#pragma unroll
    for (int e = 0; e < DD; e++) sk[e] = sk[e] + h * (acc1[e] + r[e]);
#pragma unroll
    for (int i = 0; i < D; i++) mk[i] = mk[i] + h * (pm[i] + (-y[i] + b1[i]));
  } else if (METHOD == VGPA_ODE_RK2) {
    // mean predictor uses A_k; covariance predictor uses S_k in the place of A_k (Q2)
    double pm[D], AM[DD];
    matvec<D>(A0, mk, y);
#pragma unroll
    for (int i = 0; i < D; i++) { pm[i] = -y[i] + b0[i]; xv[i] = mk[i] + h * pm[i]; }
    rhs_fwd<D>(sk, sk, sig, r);
#pragma unroll
    for (int e = 0; e < DD; e++) X[e] = sk[e] + h * r[e];
    mid_mat<D>(A0, A1, AM);
    rhs_fwd<D>(AM, X, sig, r);
    matvec<D>(AM, xv, y);
#pragma unroll
    for (int e = 0; e < DD; e++) sk[e] = sk[e] + dt * r[e];
#pragma unroll
    for (int i = 0; i < D; i++) mk[i] = mk[i] + dt * (-y[i] + 0.5 * (b0[i] + b1[i]));
  } else {  // RK4
    double acc1[DD], acc2[DD], AM[DD], k1[D], k2[D], k3[D], bmid[D];
    rhs_fwd<D>(A0, sk, sig, r);
    matvec<D>(A0, mk, y);
#pragma unroll
    for (int i = 0; i < D; i++) { k1[i] = -y[i] + b0[i]; xv[i] = mk[i] + h * k1[i]; bmid[i] = 0.5 * (b0[i] + b1[i]); }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc1[e] = r[e]; X[e] = sk[e] + h * r[e]; }
    mid_mat<D>(A0, A1, AM);
    rhs_fwd<D>(AM, X, sig, r);
    matvec<D>(AM, xv, y);
#pragma unroll
    for (int i = 0; i < D; i++) { k2[i] = -y[i] + bmid[i]; xv[i] = mk[i] + h * k2[i]; }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc2[e] = r[e]; X[e] = sk[e] + h * r[e]; }
    rhs_fwd<D>(AM, X, sig, r);
    matvec<D>(AM, xv, y);
#pragma unroll
    for (int i = 0; i < D; i++) { k3[i] = -y[i] + bmid[i]; xv[i] = mk[i] + dt * k3[i]; }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc2[e] = acc2[e] + r[e]; X[e] = sk[e] + dt * r[e]; }
    rhs_fwd<D>(A1, X, sig, r);
    matvec<D>(A1, xv, y);
#pragma unroll
    for (int e = 0; e < DD; e++) sk[e] = sk[e] + div6(dt * (acc1[e] + 2.0 * acc2[e] + r[e]));
#pragma unroll
    for (int i = 0; i < D; i++) mk[i] = mk[i] + div6(dt * (k1[i] + 2.0 * (k2[i] + k3[i]) + (-y[i] + b1[i])));
  }
}

// Forward moments, one lane per problem, streams staged through LDS in chunks of T grid points.  Step k consumes the operands of
// grid point k+1 (slot s of the chunk) and produces (m, S) of grid point k+1: the result takes the consumed slot.
template <int METHOD, int D, int T, bool OUT_T = false>      // OUT_T: the moments go to OdeArgs::msT straight from the registers
__global__ void __launch_bounds__(NTS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_fwd_lane(OdeArgs a) {
  constexpr int DD = D * D, NA = T * DD, NB = T * D;
  __shared__ double sA[ChunkMap<NA>::LDS];
  __shared__ double sB[ChunkMap<NB>::LDS];
  const int lane = threadIdx.x, prob0 = blockIdx.x * NTS;
  const int nvalid = (a.batch - prob0) < NTS ? (a.batch - prob0) : NTS;
  const bool live = lane < nvalid;
  const int prob = prob0 + (live ? lane : nvalid - 1);
  const int Np = a.Np;
  const double* A = a.A + (size_t)prob * a.strideA;
  const double* b = a.b + (size_t)prob * a.strideB;
  const double dt = a.dt;
  const size_t sS = (size_t)Np * DD, sm = (size_t)Np * D;

  double sk[DD], sig[DD], mk[D], A0[DD], b0[D];
  ld_mat<D>(a.S0, sk); ld_mat<D>(a.Sigma, sig); ld_vec<D>(a.m0, mk);
  ld_mat<D>(A, A0); ld_vec<D>(b, b0);
  // (S_t as its packed lower triangle, then m_t: D (D + 1) / 2 + D entries per grid point -- 9 instead of 12 at D = 3)
  constexpr int TRI = D * (D + 1) / 2, W = TRI + D;
  double* const outT = OUT_T ? a.msT + prob : nullptr;       // entry e of grid point t: outT[(t * W + e) * bpad]
  const size_t bp = (size_t)a.bpad;
  auto store_t = [&](int t) {
    if (live) {
#pragma unroll
      for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) outT[((size_t)t * W + tri_off(i) + j) * bp] = sk[i * D + j];
#pragma unroll
      for (int i = 0; i < D; i++) outT[((size_t)t * W + TRI + i) * bp] = mk[i];
    }
  };
  if (OUT_T) store_t(0);
  else if (live) { st_mat<D>(a.S + (size_t)prob * sS, sk); st_vec<D>(a.m + (size_t)prob * sm, mk); }

  const int nchunks = (Np - 1 + T - 1) / T;
  double pa[ChunkMap<NA>::NI], pb[ChunkMap<NB>::NI];
  const double* Au = a.A + (size_t)prob0 * a.strideA;        // wave-uniform stream bases
  const double* bu = a.b + (size_t)prob0 * a.strideB;
  double* Su = a.S + (size_t)prob0 * sS;
  double* mu = a.m + (size_t)prob0 * sm;
  const unsigned strA = (unsigned)a.strideA, strB = (unsigned)a.strideB;
  chunk_request<NA>(Au, strA, nvalid, (long)DD, (long)Np * DD, pa);
  chunk_request<NB>(bu, strB, nvalid, (long)D, (long)Np * D, pb);
  double* rowA = sA + lane * row_stride<NA>();
  double* rowB = sB + lane * row_stride<NB>();
  for (int c = 0; c < nchunks; c++) {
    chunk_to_lds<NA>(sA, pa);
    chunk_to_lds<NB>(sB, pb);
    wave_sync();
    if (c + 1 < nchunks) {      // the next chunk travels while this one is stepped through
      chunk_request<NA>(Au, strA, nvalid, (long)((c + 1) * T + 1) * DD, (long)Np * DD, pa);
      chunk_request<NB>(bu, strB, nvalid, (long)((c + 1) * T + 1) * D, (long)Np * D, pb);
    }
#pragma unroll 1
    for (int s = 0; s < T; s++) {
      if (c * T + s < Np - 1) {
        double A1[DD], b1[D];
#pragma unroll
        for (int e = 0; e < DD; e++) A1[e] = rowA[s * DD + e];
#pragma unroll
        for (int i = 0; i < D; i++) b1[i] = rowB[s * D + i];
        fwd_step<METHOD, D>(A0, A1, b0, b1, sig, dt, sk, mk);
        if (OUT_T) store_t(c * T + s + 1);
#pragma unroll
        for (int e = 0; e < DD; e++) { if (!OUT_T) rowA[s * DD + e] = sk[e]; A0[e] = A1[e]; }
#pragma unroll
        for (int i = 0; i < D; i++) { if (!OUT_T) rowB[s * D + i] = mk[i]; b0[i] = b1[i]; }
      }
    }
    wave_sync();
    if (!OUT_T) {
      chunk_flush<NA>(sA, Su, (unsigned)sS, nvalid, (long)(c * T + 1) * DD, (long)Np * DD);
      chunk_flush<NB>(sB, mu, (unsigned)sm, nvalid, (long)(c * T + 1) * D, (long)Np * D);
      wave_sync();
    }
  }
}

// jump (dE_obs) added after the step at index t1
template <int D>
__device__ __forceinline__ void load_jump(const OdeArgs& a, int prob, int t1, double (&js)[D * D], double (&jm)[D]) {
  constexpr int DD = D * D;
  if (a.js_dense) {
    ld_mat<D>(a.js_dense + ((size_t)prob * a.Np + t1) * DD, js);
    ld_vec<D>(a.jm_dense + ((size_t)prob * a.Np + t1) * D, jm);
  } else {
    const int n = a.obs_idx ? ldu(a.obs_idx, t1) : -1;           // (scalar load: see vgpa_internal.h)
    if (n >= 0) {
      ld_mat<D>(a.js_const, js);
      if (a.jmT) {
#pragma unroll
        for (int i = 0; i < D; i++) jm[i] = a.jmT[((size_t)n * D + i) * (size_t)a.bpad + prob];
      } else {
        ld_vec<D>(a.jm_sparse + ((size_t)prob * a.n_obs + n) * D, jm);
      }
    } else {
#pragma unroll
      for (int e = 0; e < DD; e++) js[e] = 0.0;
#pragma unroll
      for (int i = 0; i < D; i++) jm[i] = 0.0;
    }
  }
}

// one backward step t -> t-1: (pk, lk) in place; "t" operands (At, gst, gmt), "t-1" operands (Am, gsm, gmm), jump of t-1
template <int METHOD, int D>
__device__ __forceinline__ void bwd_step(const double (&At)[D * D], const double (&Am)[D * D], const double (&gst)[D * D],
                                         const double (&gsm)[D * D], const double (&gmt)[D], const double (&gmm)[D],
                                         const double (&js)[D * D], const double (&jm)[D], double dt, double (&pk)[D * D],
                                         double (&lk)[D]) {
#pragma clang fp contract(fast)
  constexpr int DD = D * D;
  const double h = 0.5 * dt;
  double r[DD], y[D], X[DD], xv[D];
  if (METHOD == VGPA_ODE_EULER) {
    rhs_bwd<D>(At, pk, gst, r);
    matvec<D>(At, lk, y);
#pragma unroll
    for (int e = 0; e < DD; e++) pk[e] = pk[e] - r[e] * dt + js[e];
#pragma unroll
    for (int i = 0; i < D; i++) lk[i] = lk[i] - (-gmt[i] + y[i]) * dt + jm[i];
  } else if (METHOD == VGPA_ODE_HEUN) {
    double acc1[DD], pl[D];
    rhs_bwd<D>(At, pk, gst, r);
    matvec<D>(At, lk, y);
#pragma unroll
    for (int i = 0; i < D; i++) { pl[i] = -gmt[i] + y[i]; xv[i] = lk[i] - pl[i] * dt; }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc1[e] = r[e]; X[e] = pk[e] - r[e] * dt; }
    rhs_bwd<D>(Am, X, gsm, r);
    matvec<D>(Am, xv, y);
#pragma unroll
    for (int e = 0; e < DD; e++) pk[e] = pk[e] - h * (acc1[e] + r[e]) + js[e];
#pragma unroll
    for (int i = 0; i < D; i++) lk[i] = lk[i] - h * (pl[i] + (-gmm[i] + y[i])) + jm[i];
  } else if (METHOD == VGPA_ODE_RK2) {
    double pl[D], AM[DD], gmid[DD];
    rhs_bwd<D>(At, pk, gst, r);
    matvec<D>(At, lk, y);
#pragma unroll
    for (int i = 0; i < D; i++) { pl[i] = -gmt[i] + y[i]; xv[i] = lk[i] - h * pl[i]; }
#pragma unroll
    for (int e = 0; e < DD; e++) X[e] = pk[e] - h * r[e];
    mid_mat<D>(Am, At, AM);
    mid_mat<D>(gsm, gst, gmid);
    rhs_bwd<D>(AM, X, gmid, r);
    matvec<D>(AM, xv, y);
#pragma unroll
    for (int e = 0; e < DD; e++) pk[e] = pk[e] - dt * r[e] + js[e];
#pragma unroll
    for (int i = 0; i < D; i++) lk[i] = lk[i] - dt * (-(0.5 * (gmm[i] + gmt[i])) + y[i]) + jm[i];
  } else {  // RK4
    double acc1[DD], acc2[DD], AM[DD], gmid[DD], k1[D], k2[D], k3[D], gvm[D];
    rhs_bwd<D>(At, pk, gst, r);
    matvec<D>(At, lk, y);
#pragma unroll
    for (int i = 0; i < D; i++) { k1[i] = -gmt[i] + y[i]; xv[i] = lk[i] - h * k1[i]; gvm[i] = 0.5 * (gmm[i] + gmt[i]); }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc1[e] = r[e]; X[e] = pk[e] - h * r[e]; }
    mid_mat<D>(Am, At, AM);
    mid_mat<D>(gsm, gst, gmid);
    rhs_bwd<D>(AM, X, gmid, r);
    matvec<D>(AM, xv, y);
#pragma unroll
    for (int i = 0; i < D; i++) { k2[i] = -gvm[i] + y[i]; xv[i] = lk[i] - h * k2[i]; }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc2[e] = r[e]; X[e] = pk[e] - h * r[e]; }
    rhs_bwd<D>(AM, X, gmid, r);
    matvec<D>(AM, xv, y);
#pragma unroll
    for (int i = 0; i < D; i++) { k3[i] = -gvm[i] + y[i]; xv[i] = lk[i] - dt * k3[i]; }
#pragma unroll
    for (int e = 0; e < DD; e++) { acc2[e] = acc2[e] + r[e]; X[e] = pk[e] - dt * r[e]; }
    rhs_bwd<D>(Am, X, gsm, r);
    matvec<D>(Am, xv, y);
#pragma unroll
    for (int e = 0; e < DD; e++) pk[e] = pk[e] - div6(dt * (acc1[e] + 2.0 * acc2[e] + r[e])) + js[e];
#pragma unroll
    for (int i = 0; i < D; i++) lk[i] = lk[i] - div6(dt * (k1[i] + 2.0 * (k2[i] + k3[i]) + (-gmm[i] + y[i]))) + jm[i];
  }
}

template <int METHOD, int D>
__global__ void __launch_bounds__(NTS) k_bwd_small(OdeArgs a) {
  constexpr int DD = D * D;
  const int prob = blockIdx.x * NTS + threadIdx.x;
  if (prob >= a.batch) return;
  const int Np = a.Np;
  const double* A = a.A + (size_t)prob * a.strideA;
  const double* gm = a.dEm + (size_t)prob * Np * D;
  const double* gs = a.dEs + (size_t)prob * Np * DD;
  double* lam = a.lam + (size_t)prob * Np * D;
  double* psi = a.psi + (size_t)prob * Np * DD;
  const double dt = a.dt;

  double pk[DD], lk[D];
#pragma unroll
  for (int e = 0; e < DD; e++) pk[e] = 0.0;
#pragma unroll
  for (int i = 0; i < D; i++) lk[i] = 0.0;
  st_mat<D>(psi + (size_t)(Np - 1) * DD, pk);
  st_vec<D>(lam + (size_t)(Np - 1) * D, lk);
  // "t" quantities of the current step and "m" (t-1) quantities, the latter requested one step ahead
  double At[DD], Am[DD], gst[DD], gsm[DD], gmt[D], gmm[D];
  ld_mat<D>(A + (size_t)(Np - 1) * DD, At); ld_mat<D>(gs + (size_t)(Np - 1) * DD, gst); ld_vec<D>(gm + (size_t)(Np - 1) * D, gmt);
  if (Np > 1) {
    ld_mat<D>(A + (size_t)(Np - 2) * DD, Am); ld_mat<D>(gs + (size_t)(Np - 2) * DD, gsm); ld_vec<D>(gm + (size_t)(Np - 2) * D, gmm);
  }

  for (int t = Np - 1; t > 0; t--) {
    double An[DD], gsn[DD], gmn[D];            // index t-2, for the next step
    const int tn = (t >= 2) ? t - 2 : 0;
    ld_mat<D>(A + (size_t)tn * DD, An); ld_mat<D>(gs + (size_t)tn * DD, gsn); ld_vec<D>(gm + (size_t)tn * D, gmn);
    double js[DD], jm[D];
    load_jump<D>(a, prob, t - 1, js, jm);

    bwd_step<METHOD, D>(At, Am, gst, gsm, gmt, gmm, js, jm, dt, pk, lk);
    st_mat<D>(psi + (size_t)(t - 1) * DD, pk);
    st_vec<D>(lam + (size_t)(t - 1) * D, lk);
#pragma unroll
    for (int e = 0; e < DD; e++) { At[e] = Am[e]; Am[e] = An[e]; gst[e] = gsm[e]; gsm[e] = gsn[e]; }
#pragma unroll
    for (int i = 0; i < D; i++) { gmt[i] = gmm[i]; gmm[i] = gmn[i]; }
  }
}

// ------------------------------------------------------------------------------------------------
// E_sde terms of one grid point in the form the recursion and the gradient assembly take them: dEsde_dS as a full D x D matrix,
// <df/dx> as a matrix (variational.py:263-281 adds it to A_t)
template <int MODEL, int D>
__device__ __forceinline__ void point_terms(const LaneSweepArgs& q, const double (&Av)[D * D], const double (&bv)[D], const double (&mv)[D],
                                            const double (&Sv)[D * D], double (&gs)[D * D], double (&gm)[D], double& e_t, double (&ef)[D],
                                            double (&edf)[D * D]) {
#pragma clang fp contract(fast)
  if constexpr (MODEL == VGPA_MODEL_L63) {
    EnergyL63 r;
    const double isg[3] = {q.isg[0], q.isg[1], q.isg[2]};
    energy_l63<false>(q.theta, isg, Av, bv, mv, Sv, r);
    gs[0] = r.ds[0]; gs[1] = r.ds[1]; gs[2] = r.ds[2];
    gs[3] = r.ds[1]; gs[4] = r.ds[3]; gs[5] = r.ds[4];
    gs[6] = r.ds[2]; gs[7] = r.ds[4]; gs[8] = r.ds[5];
#pragma unroll
    for (int i = 0; i < 3; i++) { gm[i] = r.dm[i]; ef[i] = r.ef[i]; }
    e_t = r.e_t;
    const double vS = q.theta[0], vR = q.theta[1], vB = q.theta[2];        // <df/dx>, lorenz_63.py:323-327
    edf[0] = -vS; edf[1] = vS; edf[2] = 0.0;
    edf[3] = vR - mv[2]; edf[4] = -1.0; edf[5] = -mv[0];
    edf[6] = mv[1]; edf[7] = mv[0]; edf[8] = -vB;
  } else {
    Energy1d r;
    energy_1d<MODEL>(q.theta[0], q.sigma1, Av[0], bv[0], mv[0], Sv[0], r);
    gs[0] = r.ds; gm[0] = r.dm; ef[0] = r.ef; edf[0] = r.edf; e_t = r.e_t;
  }
}

// The fused pass (see the head of this file).  Chunk c holds grid points [hi - T + 1, hi], hi = Np - 2 - c T; step s of the chunk
// goes from t = hi - s + 1 to t - 1 = hi - s, whose operands sit in slot T - 1 - s; gLa / gLb of t - 1 take the slots of A / b.
template <int METHOD, int MODEL, bool GRAD, int T>
__global__ void __launch_bounds__(NTS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_sweep_lane(LaneSweepArgs q) {
  constexpr int D = (MODEL == VGPA_MODEL_L63) ? 3 : 1, DD = D * D, NA = T * DD, NV = T * D;
  __shared__ double sA[ChunkMap<NA>::LDS];
  __shared__ double sB[ChunkMap<NV>::LDS];
  const OdeArgs& a = q.o;
  const int lane = threadIdx.x, prob0 = blockIdx.x * NTS;
  const int nvalid = (a.batch - prob0) < NTS ? (a.batch - prob0) : NTS;
  const bool live = lane < nvalid;
  const int prob = prob0 + (live ? lane : nvalid - 1);
  const int Np = a.Np;
  const double dt = a.dt;
  const size_t len_x = a.strideA;
  const long limA = (long)Np * DD, limV = (long)Np * D;
  // the moments come from the context's time-major array (OdeArgs::msT): one coalesced 512-byte load per entry and wave, straight
  // into registers, requested one step ahead
  constexpr int TRI = D * (D + 1) / 2, W = TRI + D;      // (packed lower triangle of S_t, then m_t: see k_fwd_lane)
  const double* const msT = a.msT + prob;
  const size_t bp = (size_t)a.bpad;
  auto load_ms = [&](int t, double (&Sv)[DD], double (&mv)[D]) {
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        const double v = msT[((size_t)t * W + tri_off(i) + j) * bp];
        Sv[i * D + j] = v; Sv[j * D + i] = v;
      }
#pragma unroll
    for (int i = 0; i < D; i++) mv[i] = msT[((size_t)t * W + TRI + i) * bp];
  };

  double Iv[DD];
#pragma unroll
  for (int e = 0; e < DD; e++) Iv[e] = q.isig[e];

  // grid point Np - 1: operands straight from HBM (once), Psi = 0, lam = 0
  double At[DD], gst[DD], gmt[D], e_t, pk[DD], lk[D];
  {
    double bt[D], mt[D], St[DD], ef[D], edf[DD];
    ld_mat<D>(a.A + (size_t)prob * a.strideA + (size_t)(Np - 1) * DD, At);
    ld_vec<D>(a.b + (size_t)prob * a.strideB + (size_t)(Np - 1) * D, bt);
    load_ms(Np - 1, St, mt);
    point_terms<MODEL, D>(q, At, bt, mt, St, gst, gmt, e_t, ef, edf);
#pragma unroll
    for (int e = 0; e < DD; e++) pk[e] = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) lk[i] = 0.0;
    if (GRAD) {
      double gA[DD], gB[D];
      grad_point<D>(At, bt, mt, St, ef, edf, pk, lk, Iv, dt, gA, gB);
      if (live) {
        st_mat<D>(q.g + (size_t)prob * len_x + (size_t)(Np - 1) * DD, gA);
        st_vec<D>(q.g + (size_t)prob * len_x + (size_t)Np * DD + (size_t)(Np - 1) * D, gB);
      }
    }
  }
  double esum = 0.0;

  const int nchunks = (Np - 1 + T - 1) / T;
  double pa[ChunkMap<NA>::NI], pb[ChunkMap<NV>::NI];
  const double* Au = a.A + (size_t)prob0 * a.strideA;        // wave-uniform stream bases
  const double* bu = a.b + (size_t)prob0 * a.strideB;
  double* gAu = q.g + (size_t)prob0 * len_x;
  double* gBu = gAu + (size_t)Np * DD;
  const unsigned strA = (unsigned)a.strideA, strB = (unsigned)a.strideB;
  {
    const long lo = (long)(Np - 2) - T + 1;
    chunk_request<NA>(Au, strA, nvalid, lo * DD, limA, pa);
    chunk_request<NV>(bu, strB, nvalid, lo * D, limV, pb);
  }
  // moments of the next TWO grid points down: requested two steps ahead (under a full chip the memory round trip is longer
  // than one step of a lone wave)
  // (Sc: the step's own grid point; the rotation and the request for three points down sit at the END of a step, so that the wait the
  //  rotation needs -- across the loop's back-edge the compiler makes it cover every load in flight -- comes a whole step of arithmetic
  //  behind the requests of the next chunk, which are issued in front of a chunk's first step)
  double Sc[DD], mc[D], Sn[DD], mn[D], Sn2[DD], mn2[D];
  load_ms(Np > 1 ? Np - 2 : 0, Sc, mc);
  load_ms(Np > 2 ? Np - 3 : 0, Sn, mn);
  load_ms(Np > 3 ? Np - 4 : 0, Sn2, mn2);
  // The jumps (dE_obs behind the step that ends at an observation) are requested a step ahead like the moments and rotated at the same
  // place, the END of a step.  A conditional load in the middle of the step made the compiler wait, at the TOP of every step, for the
  // previous step's possible jump loads before it overwrote their registers -- s_waitcnt vmcnt(0), i.e. for the moment requests issued
  // a few instructions earlier: one exposed memory round trip per grid point (7 600 of a step's 11 700 cycles).
  // The constant matrix jump comes through scalar loads, once; w = 1 behind an observation, else 0.
  double jsc[DD];
#pragma unroll
  for (int e = 0; e < DD; e++) jsc[e] = a.js_const ? lds_const(a.js_const, e) : 0.0;
  auto obs_at = [&](int t1) -> int { return (a.obs_idx && t1 >= 0) ? ldu(a.obs_idx, t1) : -1; };
  auto request_jump = [&](int n, double (&jm)[D]) {          // (n: wave-uniform)
    if (n >= 0) {
      if (a.jmT) {
#pragma unroll
        for (int i = 0; i < D; i++) jm[i] = a.jmT[((size_t)n * D + i) * (size_t)a.bpad + prob];
      } else {
        ld_vec<D>(a.jm_sparse + ((size_t)prob * a.n_obs + n) * D, jm);
      }
    } else {
#pragma unroll
      for (int i = 0; i < D; i++) jm[i] = 0.0;
    }
  };
  double jm_cur[D], jm_nxt[D];
  int n_cur = obs_at(Np - 2), n_nxt = obs_at(Np - 3), n_pre = obs_at(Np - 4);      // observation index of the step's, the next step's, ... grid point
  request_jump(n_cur, jm_cur);
  request_jump(n_nxt, jm_nxt);
  double* rowA = sA + lane * row_stride<NA>();
  double* rowB = sB + lane * row_stride<NV>();
  for (int c = 0; c < nchunks; c++) {
    const int hi = Np - 2 - c * T;
    const long lo = (long)hi - T + 1;
    chunk_to_lds<NA>(sA, pa);
    chunk_to_lds<NV>(sB, pb);
    wave_sync();
    if (c + 1 < nchunks) {      // the next chunk travels while this one is stepped through
      const long lon = lo - T;
      chunk_request<NA>(Au, strA, nvalid, lon * DD, limA, pa);
      chunk_request<NV>(bu, strB, nvalid, lon * D, limV, pb);
    }
    // (NOT unrolled: one step is ~11 KB of code -- the closed-form energy terms -- and six copies would not fit the instruction cache)
#pragma unroll 1
    for (int s = 0; s < T; s++) {
      const int idx = hi - s;               // grid point t - 1 of this step
      if (idx >= 0) {
        const int slot = T - 1 - s;
        double Am[DD], bm[D], mm[D], Sm[DD], gsm[DD], gmm[D], e_m, ef[D], edf[DD];
#pragma unroll
        for (int e = 0; e < DD; e++) { Am[e] = rowA[slot * DD + e]; Sm[e] = Sc[e]; }
#pragma unroll
        for (int i = 0; i < D; i++) { bm[i] = rowB[slot * D + i]; mm[i] = mc[i]; }
        point_terms<MODEL, D>(q, Am, bm, mm, Sm, gsm, gmm, e_m, ef, edf);
        esum += dt * (e_t + e_m) / 2.0;     // my_trapz, utilities.py:144 (interval [t-1, t])
        e_t = e_m;
        if (GRAD) {
#ifndef VGPA_LANE_JUMP_PREFETCH
#define VGPA_LANE_JUMP_PREFETCH 1
#endif
          double js[DD];
#if VGPA_LANE_JUMP_PREFETCH
          const double wj = n_cur >= 0 ? 1.0 : 0.0;
#pragma unroll
          for (int e = 0; e < DD; e++) js[e] = wj * jsc[e];
#else
          load_jump<D>(a, prob, idx, js, jm_cur);
#endif
          bwd_step<METHOD, D>(At, Am, gst, gsm, gmt, gmm, js, jm_cur, dt, pk, lk);
          double gA[DD], gB[D];
          grad_point<D>(Am, bm, mm, Sm, ef, edf, pk, lk, Iv, dt, gA, gB);
#pragma unroll
          for (int e = 0; e < DD; e++) { rowA[slot * DD + e] = gA[e]; At[e] = Am[e]; gst[e] = gsm[e]; }
#pragma unroll
          for (int i = 0; i < D; i++) { rowB[slot * D + i] = gB[i]; gmt[i] = gmm[i]; }
        }
        // everything loaded a step ago is consumed here ...
#pragma unroll
        for (int e = 0; e < DD; e++) { Sc[e] = Sn[e]; Sn[e] = Sn2[e]; }
#pragma unroll
        for (int i = 0; i < D; i++) { mc[i] = mn[i]; mn[i] = mn2[i]; }
#if VGPA_LANE_JUMP_PREFETCH
#pragma unroll
        for (int i = 0; i < D; i++) jm_cur[i] = jm_nxt[i];
        n_cur = n_nxt; n_nxt = n_pre;
#endif
        // ... and everything the steps after the next one need is requested behind it
        load_ms(idx > 2 ? idx - 3 : 0, Sn2, mn2);
#if VGPA_LANE_JUMP_PREFETCH
        request_jump(n_nxt, jm_nxt);
        n_pre = obs_at(idx - 3);
#endif
      }
    }
    if (GRAD) {
      wave_sync();
      chunk_flush<NA>(sA, gAu, (unsigned)len_x, nvalid, lo * DD, limA);
      chunk_flush<NV>(sB, gBu, (unsigned)len_x, nvalid, lo * D, limV);
    }
    wave_sync();
  }
  if (live) {
    const double esde = q.pre * esum / q.div;
    q.esde[prob] = esde;
    q.f[prob] = q.e0 + esde + q.eobs[prob];
  }
}

template <int METHOD, bool FWD, int D>
hipError_t launch_d(const OdeArgs& a, hipStream_t st) {
  dim3 grid((a.batch + NTS - 1) / NTS), block(NTS);
  constexpr int T = (D == 1) ? 16 : (D == 2 ? 8 : 4);      // grid points per chunk: the LDS of four waves per CU decides
  if (FWD && a.msT) {
    if constexpr (D == 1 || D == 3) {      // (the models of the fused lane pass)
      constexpr int TT = D == 3 ? 6 : 16;
#ifdef VGPA_LANE_T_EXPERIMENTS
      if (D == 3 && METHOD == VGPA_ODE_RK4) {
        const char* e = getenv("VGPA_LANE_T_FWD");
        const int t = e ? atoi(e) : TT;
        if (t == 4) { hipLaunchKernelGGL((k_fwd_lane<METHOD, D, 4, true>), grid, block, 0, st, a); return hipGetLastError(); }
        if (t == 8) { hipLaunchKernelGGL((k_fwd_lane<METHOD, D, 8, true>), grid, block, 0, st, a); return hipGetLastError(); }
        if (t == 12) { hipLaunchKernelGGL((k_fwd_lane<METHOD, D, 12, true>), grid, block, 0, st, a); return hipGetLastError(); }
      }
#endif
      hipLaunchKernelGGL((k_fwd_lane<METHOD, D, TT, true>), grid, block, 0, st, a);
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  if (FWD) hipLaunchKernelGGL((k_fwd_lane<METHOD, D, T>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_bwd_small<METHOD, D>), grid, block, 0, st, a);
  return hipGetLastError();
}

template <int METHOD, bool FWD>
hipError_t launch_m(const OdeArgs& a, hipStream_t st) {
  switch (a.D) {
    case 1: return launch_d<METHOD, FWD, 1>(a, st);
    case 2: return launch_d<METHOD, FWD, 2>(a, st);
    case 3: return launch_d<METHOD, FWD, 3>(a, st);
    case 4: return launch_d<METHOD, FWD, 4>(a, st);
  }
  return hipErrorInvalidValue;
}

template <int METHOD, int MODEL>
hipError_t launch_sweep_mm(const LaneSweepArgs& q, hipStream_t st) {
  dim3 grid((q.o.batch + NTS - 1) / NTS), block(NTS);
  // Lorenz-63: chunks of 4 grid points.  6 (round 4, first version) keeps 108 registers of the next chunk in flight and, with the
  // moments and jumps of the steps ahead, spills 6-19 of the wave's 512 -- and every reload of a spilled value inside the step loop is
  // a vector memory operation the compiler waits for with vmcnt(0), i.e. together with every request in flight.  Same box, T = 4 | 6:
  // 4.67 | 5.16-5.49 ms per 65536 problems (tools/lane_t_scan.sh).
  constexpr int T = (MODEL == VGPA_MODEL_L63) ? 4 : 16;
#ifdef VGPA_LANE_T_EXPERIMENTS
  if (MODEL == VGPA_MODEL_L63 && METHOD == VGPA_ODE_RK4 && q.want_grad) {
    const char* e = getenv("VGPA_LANE_T_BWD");
    const int t = e ? atoi(e) : T;
    if (t == 6) { hipLaunchKernelGGL((k_sweep_lane<METHOD, MODEL, true, 6>), grid, block, 0, st, q); return hipGetLastError(); }
    if (t == 5) { hipLaunchKernelGGL((k_sweep_lane<METHOD, MODEL, true, 5>), grid, block, 0, st, q); return hipGetLastError(); }
    if (t == 8) { hipLaunchKernelGGL((k_sweep_lane<METHOD, MODEL, true, 8>), grid, block, 0, st, q); return hipGetLastError(); }
  }
#endif
  if (q.want_grad) hipLaunchKernelGGL((k_sweep_lane<METHOD, MODEL, true, T>), grid, block, 0, st, q);
  else hipLaunchKernelGGL((k_sweep_lane<METHOD, MODEL, false, T>), grid, block, 0, st, q);
  return hipGetLastError();
}

template <int METHOD>
hipError_t launch_sweep_m(const LaneSweepArgs& q, hipStream_t st) {
  switch (q.model) {
    case VGPA_MODEL_OU: return launch_sweep_mm<METHOD, VGPA_MODEL_OU>(q, st);
    case VGPA_MODEL_DW: return launch_sweep_mm<METHOD, VGPA_MODEL_DW>(q, st);
    case VGPA_MODEL_L63: return launch_sweep_mm<METHOD, VGPA_MODEL_L63>(q, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_ode_small(int method, bool fwd, const OdeArgs& a, hipStream_t st) {
  if (a.D < 1 || a.D > kMaxLaneD) return hipErrorInvalidValue;
  switch (method) {
    case VGPA_ODE_EULER: return fwd ? launch_m<VGPA_ODE_EULER, true>(a, st) : launch_m<VGPA_ODE_EULER, false>(a, st);
    case VGPA_ODE_HEUN: return fwd ? launch_m<VGPA_ODE_HEUN, true>(a, st) : launch_m<VGPA_ODE_HEUN, false>(a, st);
    case VGPA_ODE_RK2: return fwd ? launch_m<VGPA_ODE_RK2, true>(a, st) : launch_m<VGPA_ODE_RK2, false>(a, st);
    case VGPA_ODE_RK4: return fwd ? launch_m<VGPA_ODE_RK4, true>(a, st) : launch_m<VGPA_ODE_RK4, false>(a, st);
  }
  return hipErrorInvalidValue;
}

namespace {
// msT [Np][W][bpad] (W = D (D + 1) / 2 + D: packed lower triangle of S_t, m_t) -> m [B][Np][D], S [B][Np][D][D] (both triangles): one wave per (64 problems, grid point); coalesced reads, per-lane writes
// (on demand only: vgpa_fetch and the separate kernels of the four-kernel path)
__global__ void __launch_bounds__(NTS) k_ms_untranspose(int D, int Np, int batch, int bpad, const double* __restrict__ msT,
                                                        double* __restrict__ m, double* __restrict__ S) {
  const int prob = blockIdx.y * NTS + threadIdx.x, t = blockIdx.x;      // (t in grid.x: Np may exceed the 65 535 of grid.y)
  if (prob >= batch) return;
  const int DD = D * D, TRI = D * (D + 1) / 2, W = TRI + D;
  const double* src = msT + (size_t)t * W * bpad + prob;
  double* so = S + ((size_t)prob * Np + t) * DD;
  double* mo = m + ((size_t)prob * Np + t) * D;
  for (int i = 0; i < D; i++)
    for (int j = 0; j <= i; j++) {
      const double v = src[(size_t)(tri_off(i) + j) * bpad];
      so[i * D + j] = v; so[j * D + i] = v;
    }
  for (int i = 0; i < D; i++) mo[i] = src[(size_t)(TRI + i) * bpad];
}

// Observation terms of the fused lane pass (gaussian_like.py:69-243; same expressions as assemble.hip::k_obs, one lane per problem
// looping over the observations): E_obs and the sparse vector jumps, moments read from msT, jumps written to jmT -- both with the
// problem fastest, so every access of a wave is one coalesced 512-byte transaction.  (Quirk Q4 kept: the covariance diagonal of
// observation n is taken at grid index n, not at t_n; the 1-D models use S[t_n].)
template <int D>
__global__ void __launch_bounds__(NTS) k_obs_lane(ObsArgs a, const double* __restrict__ msT, int bpad, double* __restrict__ jmT) {
  const int prob = blockIdx.x * NTS + threadIdx.x;
  if (prob >= a.batch) return;
  constexpr int TRI = D * (D + 1) / 2, W = TRI + D;
  const size_t bp = (size_t)bpad;
  const double* ms = msT + prob;
  double* jo = jmT + prob;
  const int M = a.n_obs;
  double part = 0.0;
  if (a.single) {
    const double rinv = a.Q[0], k0 = a.K[0];
    for (int n = 0; n < M; n++) {
      const size_t tn = (size_t)a.obs_t[n];
      const double y = a.obs_y[n], ss = ms[(tn * W) * bp], mm = ms[(tn * W + 1) * bp];
      const double ex2 = mm * mm + ss;
      part += (y * y) - 2.0 * y * mm + ex2;
      jo[(size_t)n * bp] = -(y - k0 * mm) * rinv;
    }
    a.eobs[prob] = 0.5 * part * rinv + a.obs_const;
    return;
  }
  for (int n = 0; n < M; n++) {
    const size_t tn = (size_t)a.obs_t[n];
    const double* y = a.obs_y + (size_t)n * D;
    double w[D];
#pragma unroll
    for (int j = 0; j < D; j++) w[j] = y[j] - ms[(tn * W + TRI + j) * bp];
#pragma unroll
    for (int i = 0; i < D; i++) {
      double qrow = 0.0, krow = 0.0;
      if (a.diag) {
        qrow = __builtin_fma(a.Q[i * D + i], w[i], qrow);
        krow = __builtin_fma(a.K[i * D + i], w[i], krow);
      } else {
#pragma unroll
        for (int j = 0; j < D; j++) {
          qrow = __builtin_fma(a.Q[i * D + j], w[j], qrow);
          krow = __builtin_fma(a.K[i * D + j], w[j], krow);
        }
      }
      jo[((size_t)n * D + i) * bp] = -krow;
      part += w[i] * qrow + a.rinv_diag[i] * ms[((size_t)n * W + tri_off(i) + i) * bp];
    }
  }
  a.eobs[prob] = 0.5 * (part + a.obs_const);
}
}  // namespace

hipError_t launch_ms_untranspose(int D, int Np, int batch, int bpad, const double* msT, double* m, double* S, hipStream_t st) {
  if ((batch + NTS - 1) / NTS > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ms_untranspose, dim3(Np, (batch + NTS - 1) / NTS), dim3(NTS), 0, st, D, Np, batch, bpad, msT, m, S);
  return hipGetLastError();
}

hipError_t launch_obs_lane(const ObsArgs& a, const double* msT, int bpad, double* jmT, hipStream_t st) {
  const dim3 grid((a.batch + NTS - 1) / NTS), block(NTS);
  if (a.D == 1) hipLaunchKernelGGL(k_obs_lane<1>, grid, block, 0, st, a, msT, bpad, jmT);
  else if (a.D == 3) hipLaunchKernelGGL(k_obs_lane<3>, grid, block, 0, st, a, msT, bpad, jmT);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

bool sweep_lane_supported(int model, int D) {
  return (model == VGPA_MODEL_L63 && D == 3) || ((model == VGPA_MODEL_OU || model == VGPA_MODEL_DW) && D == 1);
}

hipError_t launch_sweep_lane(int method, const LaneSweepArgs& q, hipStream_t st) {
  if (!sweep_lane_supported(q.model, q.o.D) || q.o.js_dense || q.o.Np < 2 || !q.o.msT) return hipErrorInvalidValue;
  switch (method) {
    case VGPA_ODE_EULER: return launch_sweep_m<VGPA_ODE_EULER>(q, st);
    case VGPA_ODE_HEUN: return launch_sweep_m<VGPA_ODE_HEUN>(q, st);
    case VGPA_ODE_RK2: return launch_sweep_m<VGPA_ODE_RK2>(q, st);
    case VGPA_ODE_RK4: return launch_sweep_m<VGPA_ODE_RK4>(q, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace vgpa
