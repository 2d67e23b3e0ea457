// Observation energy + jump terms, gradient assembly w.r.t. (A_t, b_t), trapezoid reduction -> F.
//
//   E_obs, dE_obs/dm, dE_obs/dS : src/var_bayes/gaussian_like.py:69-243 (quirk Q4 in the n-D energy)
//   gradient                    : src/var_bayes/variational.py:202-334 (_dEsde_db :332, _dEsde_da :320,
//                                 _grad_at :308)
//   E_sde = trapezoid(e_t)      : src/numerics/utilities.py:144-201 (piecewise sums == one global sum)
#include "vgpa_internal.h"

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
  const double* S = a.S + (size_t)prob * a.Np * D * D;
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
    for (int j = 0; j < D; j++) {
      const double w = y[j] - mt[j];
      qrow = __builtin_fma(a.Q[i * D + j], w, qrow);
      krow = __builtin_fma(a.K[i * D + j], w, krow);
    }
    jm[u] = -krow;
    // Q4: the covariance diagonal is taken at index n (observation counter), not at t_n
    part += (y[i] - mt[i]) * qrow + a.rinv_diag[i] * S[((size_t)n * D + i) * D + i];
  }
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
  double mv[D], rv[D], uv[D], pa[DD], q[DD];
  for (int i = 0; i < D; i++) mv[i] = a.m[o * D + i];
  for (int i = 0; i < D; i++) {
    double s = 0.0;
    for (int k = 0; k < D; k++) s = __builtin_fma(At[i * D + k], mv[k], s);
    rv[i] = -a.Ef[o * D + i] - s + bt[i];
  }
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++)
      pa[i * D + j] = (a.Edf ? a.Edf[o * DD + i * D + j] : edf_entry(a.model, a.theta, D, i, j, mv, St[0])) + At[i * D + j];
  for (int i = 0; i < D; i++) {
    double deb = 0.0;
    for (int l = 0; l < D; l++) deb = __builtin_fma(a.isig[i * D + l], rv[l], deb);
    uv[i] = deb + a.lam[o * D + i];
    for (int j = 0; j < D; j++) {
      double s = 0.0;
      for (int l = 0; l < D; l++) s = __builtin_fma(a.isig[i * D + l], pa[l * D + j], s);
      q[i * D + j] = s - 2.0 * Pt[i * D + j];
    }
  }
  const size_t len_x = (size_t)a.Np * DD + (size_t)a.Np * D;
  double* gA = a.g + (size_t)prob * len_x + (size_t)t * DD;
  double* gB = a.g + (size_t)prob * len_x + (size_t)a.Np * DD + (size_t)t * D;
  for (int i = 0; i < D; i++) {
    gB[i] = a.dt * uv[i];
    for (int j = 0; j < D; j++) {
      double s = 0.0;
      for (int k = 0; k < D; k++) s = __builtin_fma(q[i * D + k], St[k * D + j], s);
      gA[i * D + j] = a.dt * (s - uv[i] * mv[j]);
    }
  }
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

}  // namespace

hipError_t launch_obs(const ObsArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_obs, dim3(a.batch), dim3(NT), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_obs_dense(const ObsArgs& a, const double* js_const, double* jm_dense, double* js_dense,
                            hipStream_t st) {
  if (a.n_obs > 0) hipLaunchKernelGGL(k_obs_dense, dim3(a.n_obs, a.batch), dim3(NT), 0, st, a, js_const, jm_dense, js_dense);
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
