// Observation energy + jump terms, gradient assembly w.r.t. (A_t, b_t), trapezoid reduction -> F.
//
//   E_obs, dE_obs/dm, dE_obs/dS : src/var_bayes/gaussian_like.py:69-243 (quirk Q4 in the n-D energy)
//   gradient                    : src/var_bayes/variational.py:202-334 (_dEsde_db :332, _dEsde_da :320,
//                                 _grad_at :308)
//   E_sde = trapezoid(e_t)      : src/numerics/utilities.py:144-201 (piecewise sums == one global sum)
#include "vgpa_internal.h"
#include "energy_small.h"

namespace vgpa {
namespace {

constexpr int NT = 256;

__device__ __forceinline__ int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// deterministic block sum of one value per thread (fixed tree)
__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// One workgroup per problem.  jm_sparse[n][i] = -(K (y_n - m[t_n]))_i ; eobs = 0.5*(sum_n term_n + const)
__global__ void __launch_bounds__(NT) k_obs(ObsArgs a) {
  __shared__ double red[NT];
  const int D = a.D, M = a.n_obs, prob = blockIdx.x, tid = threadIdx.x;
  const double* m = a.m + (size_t)prob * a.Np * D;
  const int MS = a.s_packed ? tri_off(D) : D * D;          // doubles per stored S_t (packed lower triangle: OdeArgs::s_packed)
  const double* S = a.S + (size_t)prob * a.Np * MS;
  double* jm = a.jm_sparse + (size_t)prob * M * D;
  double part = 0.0;
  if (a.single) {
    const double rinv = a.Q[0];        // 1/r
    for (int n = tid; n < M; n += NT) {
      const int64_t tn = a.obs_t[n];
      const double y = a.obs_y[n], mm = m[tn], ss = S[tn];
      const double ex2 = mm * mm + ss;
      part += (y * y) - 2.0 * y * mm + ex2;       // gaussian_like.py:87-92 (divided by r below)
      jm[n] = -(y - a.K[0] * mm) * rinv;          // gradients_1d: -(y - H m)/r, H = 1
    }
    const double tot = block_sum(part, red);
    if (tid == 0) a.eobs[prob] = 0.5 * tot * rinv + a.obs_const;
    return;
  }
  // n-D: thread per (n, i)
  for (int u = tid; u < M * D; u += NT) {
    const int n = u / D, i = u - n * D;
    const int64_t tn = a.obs_t[n];
    const double* y = a.obs_y + (size_t)n * D;
    const double* mt = m + (size_t)tn * D;
    double qrow = 0.0, krow = 0.0;
    if (a.diag) {                                  // diagonal R^-1 and H^T R^-1 (the usual case): the other terms are exact zeros
      const double w = y[i] - mt[i];
      qrow = __builtin_fma(a.Q[i * D + i], w, qrow);
      krow = __builtin_fma(a.K[i * D + i], w, krow);
    } else {
      for (int j = 0; j < D; j++) {
        const double w = y[j] - mt[j];
        qrow = __builtin_fma(a.Q[i * D + j], w, qrow);
        krow = __builtin_fma(a.K[i * D + j], w, krow);
      }
    }
    jm[u] = -krow;
    // Q4: the covariance diagonal is taken at index n (observation counter), not at t_n
    part += (y[i] - mt[i]) * qrow + a.rinv_diag[i] * S[(size_t)n * MS + (a.s_packed ? tri_off(i) + i : i * D + i)];
  }
  const double tot = block_sum(part, red);
  if (tid == 0) a.eobs[prob] = 0.5 * (tot + a.obs_const);
}

// n-D, one workgroup per (observation, problem): the variant for large D / many observations.  part[prob][n] = term_n.
__global__ void __launch_bounds__(NT) k_obs_nd(ObsArgs a) {
  __shared__ double red[NT];
  const int D = a.D, M = a.n_obs, n = blockIdx.x, prob = blockIdx.y, tid = threadIdx.x;
  const int64_t tn = a.obs_t[n];
  const double* y = a.obs_y + (size_t)n * D;
  const double* mt = a.m + ((size_t)prob * a.Np + tn) * D;
  const double* S = a.S + (size_t)prob * a.Np * D * D;
  double* jm = a.jm_sparse + ((size_t)prob * M + n) * D;
  double part = 0.0;
  if (a.diag) {
    for (int i = tid; i < D; i += NT) {
      const double w = y[i] - mt[i];
      jm[i] = -(a.K[(size_t)i * D + i] * w);
      part += w * (a.Q[(size_t)i * D + i] * w) + a.rinv_diag[i] * S[((size_t)n * D + i) * D + i];   // Q4: S[n], not S[t_n]
    }
  } else {
    // one wave per row i: coalesced reads of row i of Q and K
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < D; i += NT / 64) {
      double qrow = 0.0, krow = 0.0;
      for (int j = lane; j < D; j += 64) {
        const double w = y[j] - mt[j];
        qrow = __builtin_fma(a.Q[(size_t)i * D + j], w, qrow);
        krow = __builtin_fma(a.K[(size_t)i * D + j], w, krow);
      }
      for (int o = 32; o > 0; o >>= 1) { qrow += __shfl_xor(qrow, o, 64); krow += __shfl_xor(krow, o, 64); }
      if (lane == 0) {
        jm[i] = -krow;
        part += (y[i] - mt[i]) * qrow + a.rinv_diag[i] * S[((size_t)n * D + i) * D + i];
      }
    }
  }
  const double tot = block_sum(part, red);
  if (tid == 0) a.part[(size_t)prob * M + n] = tot;
}

__global__ void __launch_bounds__(NT) k_obs_fin(ObsArgs a) {
  __shared__ double red[NT];
  const int M = a.n_obs, prob = blockIdx.x, tid = threadIdx.x;
  double part = 0.0;
  for (int n = tid; n < M; n += NT) part += a.part[(size_t)prob * M + n];
  const double tot = block_sum(part, red);
  if (tid == 0) a.eobs[prob] = 0.5 * (tot + a.obs_const);
}

// dense jump arrays for the operator-level API (zero off the observation rows)
__global__ void __launch_bounds__(NT) k_obs_dense(ObsArgs a, const double* js_const, double* jm_dense, double* js_dense) {
  const int D = a.D, M = a.n_obs, prob = blockIdx.y, n = blockIdx.x;
  const int64_t tn = a.obs_t[n];
  const double* jm = a.jm_sparse + ((size_t)prob * M + n) * D;
  for (int i = threadIdx.x; i < D; i += NT) jm_dense[((size_t)prob * a.Np + tn) * D + i] = jm[i];
  for (int e = threadIdx.x; e < D * D; e += NT) js_dense[((size_t)prob * a.Np + tn) * D * D + e] = js_const[e];
}

// <df/dx>[i][j] recomputed from the model (so that the dense (Np,D,D) array is never materialised)
__device__ __forceinline__ double edf_entry(int model, const double* th, int D, int i, int j, const double* mv,
                                            double s00) {
  if (model == VGPA_MODEL_L96) {
    const int ip1 = wrapi(i + 1, D), im1 = wrapi(i - 1, D), im2 = wrapi(i - 2, D);
    double v = 0.0;
    if (j == i) v = -1.0;
    if (j == ip1) v = mv[im1];
    if (j == im2) v = -mv[im1];
    if (j == im1) v = mv[ip1] - mv[im2];
    return v;
  }
  if (model == VGPA_MODEL_L63) {
    const int e = i * 3 + j;
    switch (e) {
      case 0: return -th[0];
      case 1: return th[0];
      case 2: return 0.0;
      case 3: return th[1] - mv[2];
      case 4: return -1.0;
      case 5: return -mv[0];
      case 6: return mv[1];
      case 7: return mv[0];
      default: return -th[2];
    }
  }
  if (model == VGPA_MODEL_OU) return -th[0];
  return 4.0 * (th[0] - 3.0 * (mv[0] * mv[0] + s00));
}

// One workgroup per (grid point, problem):  LDS holds Q = Sigma^-1 (Edf + A) - 2 Psi and S.
//   dEb  = Sigma^-1 (-Ef - A m + b)
//   gLa  = dt * ( Q S - (dEb + lam) m^T )          gLb = dt * (dEb + lam)
__global__ void __launch_bounds__(NT) k_grad(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int D = a.D, DD = D * D, LD = D + 1;
  const int t = blockIdx.x, prob = blockIdx.y, tid = threadIdx.x;
  const size_t o = (size_t)prob * a.Np + t;
  double* Q = smem;               // [D][LD]
  double* Ss = Q + D * LD;        // [D][LD]
  double* P = Ss + D * LD;        // [D][LD]  Edf + A (needed only for a dense Sigma^-1)
  double* mv = P + (a.sigma_diag ? 0 : D * LD);
  double* rv = mv + D;            // -Ef - A m + b
  double* uv = rv + D;            // dEb + lam
  const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * DD;
  const double* bt = a.b + (size_t)prob * a.strideB + (size_t)t * D;
  const double* St = a.S + o * DD;
  const double* Pt = a.psi + o * DD;
  const double* Edf = a.Edf ? a.Edf + o * DD : nullptr;
  const size_t len_x = (size_t)a.Np * DD + (size_t)a.Np * D;
  double* gA = a.g + (size_t)prob * len_x + (size_t)t * DD;
  double* gB = a.g + (size_t)prob * len_x + (size_t)a.Np * DD + (size_t)t * D;

  if (tid < D) mv[tid] = a.m[o * D + tid];
  __syncthreads();
  const double s00 = St[0];
  for (int e = tid; e < DD; e += NT) {
    const int i = e / D, j = e - i * D;
    const double ed = Edf ? Edf[e] : edf_entry(a.model, a.theta, D, i, j, mv, s00);
    const double pa = ed + At[e];
    Ss[i * LD + j] = St[e];
    if (a.sigma_diag) Q[i * LD + j] = a.isig[i * D + i] * pa - 2.0 * Pt[e];
    else P[i * LD + j] = pa;
  }
  if (tid < D) {
    double s = 0.0;
    for (int k = 0; k < D; k++) s = __builtin_fma(At[tid * D + k], mv[k], s);
    rv[tid] = -a.Ef[o * D + tid] - s + bt[tid];
  }
  __syncthreads();
  if (!a.sigma_diag) {
    for (int e = tid; e < DD; e += NT) {
      const int i = e / D, j = e - i * D;
      double s = 0.0;
      for (int l = 0; l < D; l++) s = __builtin_fma(a.isig[i * D + l], P[l * LD + j], s);
      Q[i * LD + j] = s - 2.0 * Pt[e];
    }
  }
  if (tid < D) {
    double deb;
    if (a.sigma_diag) deb = a.isig[tid * D + tid] * rv[tid];
    else { deb = 0.0; for (int l = 0; l < D; l++) deb = __builtin_fma(a.isig[tid * D + l], rv[l], deb); }
    const double u = deb + a.lam[o * D + tid];
    uv[tid] = u;
    gB[tid] = a.dt * u;
  }
  __syncthreads();
  for (int e = tid; e < DD; e += NT) {
    const int i = e / D, j = e - i * D;
    double s = 0.0;
    for (int k = 0; k < D; k++) s = __builtin_fma(Q[i * LD + k], Ss[k * LD + j], s);
    gA[e] = a.dt * (s - uv[i] * mv[j]);
  }
}

// ---- the same assembly with the D^3 product Q.S on the fp64 matrix cores (diagonal Sigma^-1, 5 <= D <= 64) ------------
// One workgroup of four waves per (grid point, problem).  LDS: QT[k][i] = Q[i][k] (the A-operand of
// v_mfma_f64_4x4x4_4b_f64, which contracts over rows) and Ss[k][j], both with an odd leading dimension.  Wave w owns the
// block-rows I = w, w+4, ... of the 4x4-blocked result; a "unit" (one accumulator) is block-row I x sixteen columns,
// and the left-over column blocks of REM-wide rows are packed G block-rows per unit (same scheme as the stepping
// kernels, ode_mfma_impl.h).  Operand fetches are ~1/7 of the LDS traffic of the scalar inner product, which is what
// bounded the VALU version; the kernel is left with its HBM streams (A, S, Psi in, gLa out).
template <int NB>
__global__ void __launch_bounds__(NT) k_grad_mfma(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int P = 4 * NB, LD = P + 1;
  constexpr int NQ = NB / 4, REM = NB % 4, G = REM ? 4 / REM : 0;
  constexpr int NLEFT = REM ? (NB + G - 1) / G : 0;
  constexpr int RW = (NB + 3) / 4;            // block-rows per wave
  constexpr int LW = (NLEFT + 3) / 4;         // left-over units per wave
  constexpr int rem = REM ? REM : 1;
  const int D = a.D, DD = D * D;
  const int t = blockIdx.x, prob = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const size_t o = (size_t)prob * a.Np + t;
  double* QT = smem;              // [P][LD]
  double* Ss = QT + P * LD;       // [P][LD]
  double* mv = Ss + P * LD;       // [P]
  double* rv = mv + P;            // -Ef - A m + b
  double* uv = rv + P;            // dEb + lam
  const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * DD;
  const double* bt = a.b + (size_t)prob * a.strideB + (size_t)t * D;
  const double* St = a.S + o * DD;
  const double* Pt = a.psi + o * DD;
  const double* Edf = a.Edf ? a.Edf + o * DD : nullptr;
  const size_t len_x = (size_t)a.Np * DD + (size_t)a.Np * D;
  double* gA = a.g + (size_t)prob * len_x + (size_t)t * DD;
  double* gB = a.g + (size_t)prob * len_x + (size_t)a.Np * DD + (size_t)t * D;

  // ALL HBM loads of this thread are issued before the first one is consumed -- the matrices, the diagonal of Sigma^-1
  // for its elements, and (threads < D) the vector entries: one memory round trip per workgroup instead of three
  // (with six workgroups per CU there is little else to hide a round trip behind)
  constexpr int EPT = (P * P + NT - 1) / NT;
  const unsigned magic = ((1u << 20) + (unsigned)D - 1u) / (unsigned)D;   // e / D for e < 4096, 5 <= D <= 64
  double sv[EPT], pv[EPT], av[EPT], ig[EPT], ev[EPT];
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = tid + q * NT;
    const bool in = e < DD;
    const int i = (int)(((unsigned)(in ? e : 0) * magic) >> 20);
    sv[q] = in ? St[e] : 0.0; pv[q] = in ? Pt[e] : 0.0; av[q] = (in && !a.psi_is_q) ? At[e] : 0.0;
    ig[q] = in ? a.isig[i * D + i] : 0.0;
    ev[q] = (in && Edf) ? Edf[e] : 0.0;
  }
  const double s00 = St[0];
  const bool vt = tid < D;
  const double v_m = vt ? a.m[o * D + tid] : 0.0;
  const double v_am = (vt && a.Am) ? a.Am[o * D + tid] : 0.0;
  const double v_ef = vt ? a.Ef[o * D + tid] : 0.0;
  const double v_b = vt ? bt[tid] : 0.0;
  const double v_ig = vt ? a.isig[tid * D + tid] : 0.0;
  const double v_lam = vt ? a.lam[o * D + tid] : 0.0;
  const bool l96_band = !Edf && a.model == VGPA_MODEL_L96;
  if (tid < P) { mv[tid] = v_m; uv[tid] = 0.0; }
  if (D < P) {                                   // zero padding of the operands (rows / columns D..P-1)
    for (int e = tid; e < P * LD; e += NT) { QT[e] = 0.0; Ss[e] = 0.0; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = tid + q * NT;
    if (e < DD) {
      const int i = (int)(((unsigned)e * magic) >> 20), j = e - i * D;
      // Lorenz-96 (the only model with D >= 5): the four-entry band of <df/dx> is added by the row's thread behind the barrier, so
      // that this pass has no predicates (round 3: -0.5 ms per 512-problem launch)
      const double ed = Edf ? ev[q] : (l96_band ? 0.0 : edf_entry(a.model, a.theta, D, i, j, mv, s00));
      Ss[i * LD + j] = sv[q];
      // (psi_is_q: the backward kernel left Q''_t = Sigma^-1 A_t - 2 Psi_t where Psi_t would be -- one stream less)
      QT[j * LD + i] = a.psi_is_q ? __builtin_fma(ig[q], ed, pv[q]) : ig[q] * (ed + av[q]) - 2.0 * pv[q];
    }
  }
  if (vt) {
    double s = v_am;
    if (!a.Am) { s = 0.0; for (int k = 0; k < D; k++) s = __builtin_fma(At[tid * D + k], mv[k], s); }
    const double r = -v_ef - s + v_b;
    const double u = v_ig * r + v_lam;
    uv[tid] = u;
    gB[tid] = a.dt * u;
  }
  __syncthreads();
  if (l96_band) {
    if (vt) {                                      // row k = tid of Sigma^-1 <df/dx> (lorenz_96.py:35-83), D >= 5: four distinct entries
      const int k = tid, kp1 = wrapi(k + 1, D), km1 = wrapi(k - 1, D), km2 = wrapi(k - 2, D);
      const double mm1 = mv[km1];
      QT[k * LD + k] -= v_ig;
      QT[kp1 * LD + k] = __builtin_fma(v_ig, mm1, QT[kp1 * LD + k]);
      QT[km2 * LD + k] = __builtin_fma(-v_ig, mm1, QT[km2 * LD + k]);
      QT[km1 * LD + k] = __builtin_fma(v_ig, mv[kp1] - mv[km2], QT[km1 * LD + k]);
    }
    __syncthreads();
  }

  // ---- Q.S on the matrix cores: acc[q][ii] = block-row (wave + 4 ii), column group q
  const int r4 = lane >> 4, c4 = lane & 3, b = (lane >> 2) & 3;
  const double* pa = QT + r4 * LD + c4;
  const double* pb = Ss + r4 * LD;
  const int colq = lane & 15;
  const int coll = 4 * (4 * NQ + b % rem) + c4;
  double acc[(NQ > 0 ? NQ : 1) * RW], accl[LW > 0 ? LW : 1];
#pragma unroll
  for (int u = 0; u < (NQ > 0 ? NQ : 1) * RW; u++) acc[u] = 0.0;
#pragma unroll
  for (int u = 0; u < (LW > 0 ? LW : 1); u++) accl[u] = 0.0;
  int rowoff[RW], leftoff[LW > 0 ? LW : 1];
#pragma unroll
  for (int ii = 0; ii < RW; ii++) { const int I = wave + 4 * ii; rowoff[ii] = 4 * (I < NB ? I : NB - 1); }
#pragma unroll
  for (int vv = 0; vv < LW; vv++) {
    const int v = wave + 4 * vv;
    int ib = (v < NLEFT ? v : NLEFT - 1) * G + b / rem;
    leftoff[vv] = 4 * (ib < NB ? ib : NB - 1);            // spare block slots read valid memory; result unused
  }
#pragma unroll
  for (int kk = 0; kk < NB; kk++) {
    double bq[NQ > 0 ? NQ : 1];
#pragma unroll
    for (int q = 0; q < NQ; q++) bq[q] = pb[kk * 4 * LD + 16 * q + colq];
    double bl = 0.0;
    if (NLEFT > 0) bl = pb[kk * 4 * LD + coll];
#pragma unroll
    for (int ii = 0; ii < RW; ii++) {
      if (NQ > 0) {
        const double af = pa[kk * 4 * LD + rowoff[ii]];
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[q * RW + ii] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bq[q], acc[q * RW + ii], 0, 0, 0);
      }
    }
#pragma unroll
    for (int vv = 0; vv < LW; vv++) {
      const double af = pa[kk * 4 * LD + leftoff[vv]];
      accl[vv] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bl, accl[vv], 0, 0, 0);
    }
  }
  // ---- gLa = dt (Q S - u m^T), straight from the accumulators (16 contiguous doubles per lane group)
#pragma unroll
  for (int ii = 0; ii < RW; ii++) {
    const int I = wave + 4 * ii;
    const int row = 4 * I + r4;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int col = 16 * q + 4 * b + c4;
      if (I < NB && row < D && col < D) gA[row * D + col] = a.dt * (acc[q * RW + ii] - uv[row] * mv[col]);
    }
  }
#pragma unroll
  for (int vv = 0; vv < LW; vv++) {
    const int v = wave + 4 * vv;
    const int Ib = v * G + b / rem;
    const int row = 4 * Ib + r4, col = 4 * (4 * NQ + b % rem) + c4;
    if (v < NLEFT && b < G * REM && Ib < NB && row < D && col < D) gA[row * D + col] = a.dt * (accl[vv] - uv[row] * mv[col]);
  }
}


#ifndef VGPA_GRAD_TPW
#define VGPA_GRAD_TPW 4
#endif
constexpr int kGradTPW = VGPA_GRAD_TPW;      // consecutive grid points per workgroup of k_grad_mfma: the loads of point t+1 are in flight under the product of t
// The same assembly for GradArgs::psi_is_q (the batched fused sweeps of 33 <= D <= 40: `psi` holds Q''_t, A_t and a dense <df/dx> are not
// streamed), software-pipelined over kGradTPW consecutive grid points of one problem.
template <int NB>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) k_grad_mfma_q(GradArgs a) {
  constexpr bool QMODE = true;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int P = 4 * NB, LD = P + 1;
  constexpr int NQ = NB / 4, REM = NB % 4, G = REM ? 4 / REM : 0;
  constexpr int NLEFT = REM ? (NB + G - 1) / G : 0;
  constexpr int RW = (NB + 3) / 4;            // block-rows per wave
  constexpr int LW = (NLEFT + 3) / 4;         // left-over units per wave
  constexpr int rem = REM ? REM : 1;
  const int D = a.D, DD = D * D;
  const int prob = blockIdx.y, tid = threadIdx.x;
  const int t_first = blockIdx.x * kGradTPW;
  const int t_end = t_first + kGradTPW < a.Np ? t_first + kGradTPW : a.Np;
  const int lane = tid & 63, wave = tid >> 6;
  double* QT = smem;              // [P][LD]
  double* Ss = QT + P * LD;       // [P][LD]
  double* mv = Ss + P * LD;       // [P]
  double* rv = mv + P;            // -Ef - A m + b
  double* uv = rv + P;            // dEb + lam
  const size_t len_x = (size_t)a.Np * DD + (size_t)a.Np * D;

  // ALL HBM loads of a grid point are issued before the first one is consumed -- the matrices and (threads < D) the vector
  // entries -- and, from the second grid point of the workgroup on, a whole product ahead of their use (round 3: the kernel is
  // bound by its HBM streams, and with five workgroups per CU the load phase of one was not covered by the others)
  constexpr int EPT = (P * P + NT - 1) / NT;
  const unsigned magic = ((1u << 20) + (unsigned)D - 1u) / (unsigned)D;   // e / D for e < 4096, 5 <= D <= 64
  double sv[EPT], pv[EPT], av[QMODE ? 1 : EPT], ig[QMODE ? 1 : EPT], ev[QMODE ? 1 : EPT];
  double v_m = 0.0, v_am = 0.0, v_ef = 0.0, v_b = 0.0, v_lam = 0.0;
  const bool vt = tid < D;
  const double v_ig = vt ? a.isig[tid * D + tid] : 0.0;
  if constexpr (!QMODE) {
#pragma unroll
    for (int q = 0; q < EPT; q++) {               // (the diagonal of Sigma^-1 of this thread's elements: the same for every grid point)
      const int e = tid + q * NT;
      const int i = (int)(((unsigned)(e < DD ? e : 0) * magic) >> 20);
      ig[q] = e < DD ? a.isig[i * D + i] : 0.0;
    }
  }
  // packed S_t (GradArgs::s_packed): this thread's flat indices e = tid + q NT of the lower triangle and their (row, column)
  const bool spk = a.s_packed != 0;
  const int PK = tri_off(D);
  int pi[EPT], pj[EPT];
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = tid + q * NT;
    pi[q] = tri_row(e < PK ? e : 0);
    pj[q] = (e < PK ? e : 0) - tri_off(pi[q]);
  }
  auto request = [&](int t) {
    const size_t o = (size_t)prob * a.Np + t;
    const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * DD;
    const double* St = a.S + o * (spk ? PK : DD);
    const double* Pt = a.psi + o * DD;
    const double* Edf = a.Edf ? a.Edf + o * DD : nullptr;
    (void)At; (void)Edf;
    const int ns = spk ? PK : DD;
#pragma unroll
    for (int q = 0; q < EPT; q++) {
      const int e = tid + q * NT;
      const bool in = e < DD;
      sv[q] = e < ns ? St[e] : 0.0; pv[q] = in ? Pt[e] : 0.0;
      if constexpr (!QMODE) { av[q] = in ? At[e] : 0.0; ev[q] = (in && Edf) ? Edf[e] : 0.0; }
    }
    v_m = vt ? a.m[o * D + tid] : 0.0;
    v_am = (vt && a.Am) ? a.Am[o * D + tid] : 0.0;
    v_ef = vt ? a.Ef[o * D + tid] : 0.0;
    v_b = vt ? a.b[(size_t)prob * a.strideB + (size_t)t * D + tid] : 0.0;
    v_lam = vt ? a.lam[o * D + tid] : 0.0;
  };
  request(t_first);
  if (tid < P) uv[tid] = 0.0;
  if (D < P) {                                   // zero padding of the operands (rows / columns D..P-1)
    for (int e = tid; e < P * LD; e += NT) { QT[e] = 0.0; Ss[e] = 0.0; }
  }

  const int r4 = lane >> 4, c4 = lane & 3, b = (lane >> 2) & 3;
  const double* pa = QT + r4 * LD + c4;
  const double* pb = Ss + r4 * LD;
  const int colq = lane & 15;
  const int coll = 4 * (4 * NQ + b % rem) + c4;
  int rowoff[RW], leftoff[LW > 0 ? LW : 1];
#pragma unroll
  for (int ii = 0; ii < RW; ii++) { const int I = wave + 4 * ii; rowoff[ii] = 4 * (I < NB ? I : NB - 1); }
#pragma unroll
  for (int vv = 0; vv < LW; vv++) {
    const int v = wave + 4 * vv;
    int ib = (v < NLEFT ? v : NLEFT - 1) * G + b / rem;
    leftoff[vv] = 4 * (ib < NB ? ib : NB - 1);            // spare block slots read valid memory; result unused
  }

#pragma unroll 1
  for (int t = t_first; t < t_end; t++) {
    const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * DD;
    double* gA = a.g + (size_t)prob * len_x + (size_t)t * DD;
    double* gB = a.g + (size_t)prob * len_x + (size_t)a.Np * DD + (size_t)t * D;
    __syncthreads();                               // the previous point's product has read the operands (first pass: the padding is written)
    if (tid < P) mv[tid] = v_m;
    __syncthreads();
    const bool has_edf = a.Edf != nullptr;
    // the operands: S_t as it is, Q^T without the <df/dx> band (one predicate-free pass; the band of the Lorenz-96 Jacobian -- four
    // entries per row -- is added by the row's thread behind the barrier)
#pragma unroll
    for (int q = 0; q < EPT; q++) {
      const int e = tid + q * NT;
      if (spk && e < PK) { Ss[pi[q] * LD + pj[q]] = sv[q]; Ss[pj[q] * LD + pi[q]] = sv[q]; }      // both triangles from the packed lower one
      if (e < DD) {
        const int i = (int)(((unsigned)e * magic) >> 20), j = e - i * D;
        if (!spk) Ss[i * LD + j] = sv[q];
        // (QMODE: the backward kernel left Q''_t = Sigma^-1 A_t - 2 Psi_t where Psi_t would be -- one stream less)
        if constexpr (QMODE) QT[j * LD + i] = pv[q];
        else QT[j * LD + i] = __builtin_fma(ig[q], has_edf ? ev[q] + av[q] : av[q], -2.0 * pv[q]);
      }
    }
    if (vt) {
      double s = v_am;
      if (!a.Am) { s = 0.0; for (int k = 0; k < D; k++) s = __builtin_fma(At[tid * D + k], mv[k], s); }
      const double r = -v_ef - s + v_b;
      const double u = v_ig * r + v_lam;
      uv[tid] = u;
      gB[tid] = a.dt * u;
    }
    __syncthreads();
    if (vt && !has_edf && a.model == VGPA_MODEL_L96) {         // row k = tid of Sigma^-1 <df/dx> (lorenz_96.py:35-83), D >= 5: four distinct entries
      const int k = tid, kp1 = wrapi(k + 1, D), km1 = wrapi(k - 1, D), km2 = wrapi(k - 2, D);
      const double mm1 = mv[km1];
      QT[k * LD + k] -= v_ig;
      QT[kp1 * LD + k] = __builtin_fma(v_ig, mm1, QT[kp1 * LD + k]);
      QT[km2 * LD + k] = __builtin_fma(-v_ig, mm1, QT[km2 * LD + k]);
      QT[km1 * LD + k] = __builtin_fma(v_ig, mv[kp1] - mv[km2], QT[km1 * LD + k]);
    }
    __syncthreads();
    if (QMODE && t + 1 < t_end) request(t + 1);   // in flight under the product below (QMODE: two matrices = 28 registers; else requested behind the product)

    // ---- Q.S on the matrix cores: acc[q][ii] = block-row (wave + 4 ii), column group q
    double acc[(NQ > 0 ? NQ : 1) * RW], accl[LW > 0 ? LW : 1];
#pragma unroll
    for (int u = 0; u < (NQ > 0 ? NQ : 1) * RW; u++) acc[u] = 0.0;
#pragma unroll
    for (int u = 0; u < (LW > 0 ? LW : 1); u++) accl[u] = 0.0;
#pragma unroll
    for (int kk = 0; kk < NB; kk++) {
      double bq[NQ > 0 ? NQ : 1];
#pragma unroll
      for (int q = 0; q < NQ; q++) bq[q] = pb[kk * 4 * LD + 16 * q + colq];
      double bl = 0.0;
      if (NLEFT > 0) bl = pb[kk * 4 * LD + coll];
#pragma unroll
      for (int ii = 0; ii < RW; ii++) {
        if (NQ > 0) {
          const double af = pa[kk * 4 * LD + rowoff[ii]];
#pragma unroll
          for (int q = 0; q < NQ; q++) acc[q * RW + ii] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bq[q], acc[q * RW + ii], 0, 0, 0);
        }
      }
#pragma unroll
      for (int vv = 0; vv < LW; vv++) {
        const double af = pa[kk * 4 * LD + leftoff[vv]];
        accl[vv] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bl, accl[vv], 0, 0, 0);
      }
      // (the next grid point's operands are in flight in registers: keep the scheduler from hoisting every fragment read of the product)
      if (kk % 3 == 2) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- gLa = dt (Q S - u m^T): through LDS (the operand buffer of Q^T is dead behind the product) and out as whole 128-byte lines.
    // Straight from the accumulators a 320-byte row is written as 128 + 128 + 64 bytes by up to three store instructions of two
    // waves; odd rows start in the middle of a line.  The partial lines do not always meet in L2 before they are evicted: the
    // kernel's WRITE_SIZE was 17 % above the bytes of its two outputs (profiles/r04r_pmc_fetch_write_B512.csv).
#ifndef VGPA_GRAD_STAGE_OUT
#define VGPA_GRAD_STAGE_OUT 1
#endif
    double* const outb = VGPA_GRAD_STAGE_OUT ? QT : gA;
    const int ldo = VGPA_GRAD_STAGE_OUT ? LD : D;      // (in LDS with the operand's leading dimension: its zero padding, D < P, stays untouched)
    if (VGPA_GRAD_STAGE_OUT) __syncthreads();      // every wave has read its fragments of Q^T
#pragma unroll
    for (int ii = 0; ii < RW; ii++) {
      const int I = wave + 4 * ii;
      const int row = 4 * I + r4;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int col = 16 * q + 4 * b + c4;
        if (I < NB && row < D && col < D) outb[row * ldo + col] = a.dt * (acc[q * RW + ii] - uv[row] * mv[col]);
      }
    }
#pragma unroll
    for (int vv = 0; vv < LW; vv++) {
      const int v = wave + 4 * vv;
      const int Ib = v * G + b / rem;
      const int row = 4 * Ib + r4, col = 4 * (4 * NQ + b % rem) + c4;
      if (v < NLEFT && b < G * REM && Ib < NB && row < D && col < D) outb[row * ldo + col] = a.dt * (accl[vv] - uv[row] * mv[col]);
    }
    if (VGPA_GRAD_STAGE_OUT) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < EPT; q++) {
        const int e = tid + q * NT;
        if (e < DD) {
          const int i = (int)(((unsigned)e * magic) >> 20);
          gA[e] = QT[i * LD + (e - i * D)];
        }
      }
    }
    if (!QMODE && t + 1 < t_end) request(t + 1);
  }
}

// small D (1..4): one thread per grid point, everything in registers
template <int D>
__global__ void __launch_bounds__(64) k_grad_small(GradArgs a) {
  const int t = blockIdx.x * 64 + threadIdx.x, prob = blockIdx.y;
  if (t >= a.Np) return;
  constexpr int DD = D * D;
  const size_t o = (size_t)prob * a.Np + t;
  const double* At = a.A + (size_t)prob * a.strideA + (size_t)t * DD;
  const double* bt = a.b + (size_t)prob * a.strideB + (size_t)t * D;
  const double* St = a.S + o * DD;
  const double* Pt = a.psi + o * DD;
  // every operand of the grid point is requested before the first one is used (one memory round trip per thread)
  double Av[DD], Sv[DD], Pv[DD], Ev[DD], Iv[DD], mv[D], bv[D], ef[D], lm[D];
#pragma unroll
  for (int e = 0; e < DD; e++) {
    Av[e] = At[e]; Sv[e] = St[e]; Pv[e] = Pt[e]; Iv[e] = a.isig[e];
    Ev[e] = a.Edf ? a.Edf[o * DD + e] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < D; i++) { mv[i] = a.m[o * D + i]; bv[i] = bt[i]; ef[i] = a.Ef[o * D + i]; lm[i] = a.lam[o * D + i]; }
  if (!a.Edf) {
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
      for (int j = 0; j < D; j++) Ev[i * D + j] = edf_entry(a.model, a.theta, D, i, j, mv, Sv[0]);
  }
  double gAv[DD], gBv[D];
  grad_point<D>(Av, bv, mv, Sv, ef, Ev, Pv, lm, Iv, a.dt, gAv, gBv);
  const size_t len_x = (size_t)a.Np * DD + (size_t)a.Np * D;
  double* gA = a.g + (size_t)prob * len_x + (size_t)t * DD;
  double* gB = a.g + (size_t)prob * len_x + (size_t)a.Np * DD + (size_t)t * D;
#pragma unroll
  for (int i = 0; i < D; i++) gB[i] = gBv[i];
#pragma unroll
  for (int e = 0; e < DD; e++) gA[e] = gAv[e];
}

// E_sde = pre * trapz(e_t, dt) / div ;  F = e0 + E_sde + E_obs.   One workgroup per problem.
__global__ void __launch_bounds__(NT) k_reduce(ReduceArgs a) {
  __shared__ double red[NT];
  const int prob = blockIdx.x, tid = threadIdx.x;
  const double* e = a.e_t + (size_t)prob * a.Np;
  double part = 0.0;
  for (int i = tid; i < a.Np - 1; i += NT) part += a.dt * (e[i + 1] + e[i]) / 2.0;
  const double tot = block_sum(part, red);
  if (tid == 0) {
    const double esde = a.pre * tot / a.div;
    a.esde[prob] = esde;
    a.f[prob] = a.e0 + esde + a.eobs[prob];
  }
}

// out[prob][h] = sum_i dt (e[i+1][h] + e[i][h]) / 2 for the H interleaved integrands of one problem (grid = (H, batch))
__global__ void __launch_bounds__(NT) k_trapz_multi(const double* e, int Np, int H, double dt, double* out) {
  __shared__ double red[NT];
  const int h = blockIdx.x, prob = blockIdx.y, tid = threadIdx.x;
  const double* ep = e + (size_t)prob * Np * H + h;
  double part = 0.0;
  for (int i = tid; i < Np - 1; i += NT) part += dt * (ep[(size_t)(i + 1) * H] + ep[(size_t)i * H]) / 2.0;
  const double tot = block_sum(part, red);
  if (tid == 0) out[(size_t)prob * H + h] = tot;
}

}  // namespace

hipError_t launch_trapz_multi(const double* e, int Np, int H, int batch, double dt, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_trapz_multi, dim3(H, batch), dim3(NT), 0, st, e, Np, H, dt, out);
  return hipGetLastError();
}

hipError_t launch_obs(const ObsArgs& a, hipStream_t st) {
  if (a.part && !a.single && a.n_obs > 0) {
    hipLaunchKernelGGL(k_obs_nd, dim3(a.n_obs, a.batch), dim3(NT), 0, st, a);
    hipLaunchKernelGGL(k_obs_fin, dim3(a.batch), dim3(NT), 0, st, a);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_obs, dim3(a.batch), dim3(NT), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_obs_dense(const ObsArgs& a, const double* js_const, double* jm_dense, double* js_dense,
                            hipStream_t st) {
  if (a.n_obs > 0) hipLaunchKernelGGL(k_obs_dense, dim3(a.n_obs, a.batch), dim3(NT), 0, st, a, js_const, jm_dense, js_dense);
  return hipGetLastError();
}

// Psi_t back from Q''_t = diag(isg) A_t - 2 Psi_t (the fused batched sweeps, OdeArgs::q_isg), in place: VGPA_FETCH_PSIT
__global__ void __launch_bounds__(NT) k_psi_from_q(int Np, int D, size_t strideA, const double* A, const double* isg, double* pq) {
  const int t = blockIdx.x, prob = blockIdx.y, DD = D * D;
  const double* At = A + (size_t)prob * strideA + (size_t)t * DD;
  double* q = pq + ((size_t)prob * Np + t) * DD;
  for (int e = threadIdx.x; e < DD; e += NT) {
    const int i = e / D;
    q[e] = 0.5 * __builtin_fma(isg[i], At[e], -q[e]);
  }
}

__global__ void __launch_bounds__(NT) k_mirror_upper(int D, double* m) {
  double* q = m + (size_t)blockIdx.x * D * D;
  for (int e = threadIdx.x; e < D * D; e += NT) {
    const int i = e / D, j = e - i * D;
    if (i > j) q[e] = q[j * D + i];
  }
}

// packed lower triangle -> full symmetric matrix, one workgroup per matrix (vgpa_fetch and every consumer that wants S_t whole)
__global__ void __launch_bounds__(NT) k_unpack_lower(int D, const double* __restrict__ packed, double* __restrict__ full) {
  const double* src = packed + (size_t)blockIdx.x * tri_off(D);
  double* dst = full + (size_t)blockIdx.x * D * D;
  for (int e = threadIdx.x; e < D * D; e += NT) {
    const int i = e / D, j = e - i * D;
    dst[e] = src[tri_idx(i, j)];
  }
}

hipError_t launch_unpack_lower(size_t n_mat, int D, const double* packed, double* full, hipStream_t st) {
  hipLaunchKernelGGL(k_unpack_lower, dim3((unsigned)n_mat), dim3(NT), 0, st, D, packed, full);
  return hipGetLastError();
}

hipError_t launch_mirror_upper(size_t n_mat, int D, double* m, hipStream_t st) {
  hipLaunchKernelGGL(k_mirror_upper, dim3((unsigned)n_mat), dim3(NT), 0, st, D, m);
  return hipGetLastError();
}

hipError_t launch_psi_from_q(int batch, int Np, int D, size_t strideA, const double* A, const double* isg, double* psi_q, hipStream_t st) {
  hipLaunchKernelGGL(k_psi_from_q, dim3(Np, batch), dim3(NT), 0, st, Np, D, strideA, A, isg, psi_q);
  return hipGetLastError();
}

hipError_t launch_grad(const GradArgs& a, hipStream_t st) {
  if (a.D <= 4) {
    dim3 grid((a.Np + 63) / 64, a.batch);
    switch (a.D) {
      case 1: hipLaunchKernelGGL(k_grad_small<1>, grid, dim3(64), 0, st, a); break;
      case 2: hipLaunchKernelGGL(k_grad_small<2>, grid, dim3(64), 0, st, a); break;
      case 3: hipLaunchKernelGGL(k_grad_small<3>, grid, dim3(64), 0, st, a); break;
      default: hipLaunchKernelGGL(k_grad_small<4>, grid, dim3(64), 0, st, a); break;
    }
    return hipGetLastError();
  }
  if (a.D > kMaxSmallD) return hipErrorInvalidValue;
  if (a.sigma_diag && !a.scalar_product) {
    const int nb = (a.D + 3) / 4, P = 4 * nb;
    const size_t lds = sizeof(double) * (size_t)(2 * P * (P + 1) + 3 * P);
#define VGPA_GRAD_CASE(NBV)                                                                                          \
  case NBV:                                                                                                          \
    if (lds > 48 * 1024)                                                                                             \
      (void)hipFuncSetAttribute((const void*)k_grad_mfma<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k_grad_mfma<NBV>, dim3(a.Np, a.batch), dim3(NT), lds, st, a);                                  \
    return hipGetLastError();
    if (a.psi_is_q && !a.Edf && (nb == 9 || nb == 10)) {          // (what sym::stores_q allows)
      if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)k_grad_mfma_q<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_grad_mfma_q<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }
      const dim3 grid((a.Np + kGradTPW - 1) / kGradTPW, a.batch);
      if (nb == 9) hipLaunchKernelGGL(k_grad_mfma_q<9>, grid, dim3(NT), lds, st, a);
      else hipLaunchKernelGGL(k_grad_mfma_q<10>, grid, dim3(NT), lds, st, a);
      return hipGetLastError();
    }
    switch (nb) {
      VGPA_GRAD_CASE(2) VGPA_GRAD_CASE(3) VGPA_GRAD_CASE(4) VGPA_GRAD_CASE(5) VGPA_GRAD_CASE(6) VGPA_GRAD_CASE(7)
      VGPA_GRAD_CASE(8) VGPA_GRAD_CASE(9) VGPA_GRAD_CASE(10) VGPA_GRAD_CASE(11) VGPA_GRAD_CASE(12)
      VGPA_GRAD_CASE(13) VGPA_GRAD_CASE(14) VGPA_GRAD_CASE(15) VGPA_GRAD_CASE(16)
      default: break;
    }
#undef VGPA_GRAD_CASE
  }
  const int LD = a.D + 1;
  const size_t lds = sizeof(double) * (size_t)((a.sigma_diag ? 2 : 3) * a.D * LD + 3 * a.D);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)k_grad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_grad, dim3(a.Np, a.batch), dim3(NT), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce, dim3(a.batch), dim3(NT), 0, st, a);
  return hipGetLastError();
}

}  // namespace vgpa
