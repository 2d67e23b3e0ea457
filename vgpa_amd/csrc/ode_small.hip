// Time stepping for D <= 4 (Ornstein-Uhlenbeck, double well: D = 1; Lorenz-63: D = 3): ONE LANE per problem, the
// whole state in registers, no LDS, no barriers.  A batch of independent problems fills the lanes; the recursion of a
// problem is sequential in t, so its latency is one lane's instruction latency (comparable to the workgroup-per-problem
// kernel of ode_generic.hip at D = 3), but 64 problems share a wave instead of each occupying a workgroup of four
// waves that synchronises six times per step: BASELINE configs[1] (Lorenz-63, RK4, Np = 1001), 65536 problems:
// forward + backward 236 ms -> see DESIGN.md s.4.2.
//
// Same arithmetic, in the same order, as ode_generic.hip (which restates the reference):
//   forward : euler.py:27-92, heun.py:28-111, runge_kutta2.py:25-102 (incl. quirk Q2 at :96), runge_kutta4.py:25-113
//   backward: euler.py:94-154, heun.py:113-190, runge_kutta2.py:104-194, runge_kutta4.py:115-211
//             (jumps added after the step, Q9; f_lam uses A.lam, Q3)
//   RHS     : f_S = -A S - S A^T + Sigma (ode_solver.py:60), f_Psi = -G + Psi A + A^T Psi (ode_solver.py:94)
// A_{t+-1}, b / dE of the next step are requested one step ahead.
#include "vgpa_internal.h"

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

// ------------------------------------------------------------------------------------------------
template <int METHOD, int D>
__global__ void __launch_bounds__(NTS) k_fwd_small(OdeArgs a) {
  constexpr int DD = D * D;
  const int prob = blockIdx.x * NTS + threadIdx.x;
  if (prob >= a.batch) return;
  const int Np = a.Np;
  const double* A = a.A + (size_t)prob * a.strideA;
  const double* b = a.b + (size_t)prob * a.strideB;
  double* mt = a.m + (size_t)prob * Np * D;
  double* st = a.S + (size_t)prob * Np * DD;
  const double dt = a.dt, h = 0.5 * a.dt;

  double sk[DD], sig[DD], mk[D];
  ld_mat<D>(a.S0, sk); ld_mat<D>(a.Sigma, sig); ld_vec<D>(a.m0, mk);
  st_mat<D>(st, sk); st_vec<D>(mt, mk);
  double A0[DD], A1[DD], b0[D], b1[D];
  ld_mat<D>(A, A0); ld_vec<D>(b, b0);
  if (Np > 1) { ld_mat<D>(A + DD, A1); ld_vec<D>(b + D, b1); }

  for (int k = 0; k < Np - 1; k++) {
    // A_{k+2}, b_{k+2} for the next step (clamped at the end of the grid; the value is then unused)
    double A2[DD], b2[D];
    const int kn = (k + 2 < Np) ? k + 2 : Np - 1;
    ld_mat<D>(A + (size_t)kn * DD, A2); ld_vec<D>(b + (size_t)kn * D, b2);

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
      for (int e = 0; e < DD; e++) sk[e] = sk[e] + dt * (acc1[e] + 2.0 * acc2[e] + r[e]) / 6.0;
#pragma unroll
      for (int i = 0; i < D; i++) mk[i] = mk[i] + dt * (k1[i] + 2.0 * (k2[i] + k3[i]) + (-y[i] + b1[i])) / 6.0;
    }
    st_mat<D>(st + (size_t)(k + 1) * DD, sk);
    st_vec<D>(mt + (size_t)(k + 1) * D, mk);
#pragma unroll
    for (int e = 0; e < DD; e++) { A0[e] = A1[e]; A1[e] = A2[e]; }
#pragma unroll
    for (int i = 0; i < D; i++) { b0[i] = b1[i]; b1[i] = b2[i]; }
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
      ld_vec<D>(a.jm_sparse + ((size_t)prob * a.n_obs + n) * D, jm);
    } else {
#pragma unroll
      for (int e = 0; e < DD; e++) js[e] = 0.0;
#pragma unroll
      for (int i = 0; i < D; i++) jm[i] = 0.0;
    }
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
  const double dt = a.dt, h = 0.5 * a.dt;

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
      for (int e = 0; e < DD; e++) pk[e] = pk[e] - dt * (acc1[e] + 2.0 * acc2[e] + r[e]) / 6.0 + js[e];
#pragma unroll
      for (int i = 0; i < D; i++) lk[i] = lk[i] - dt * (k1[i] + 2.0 * (k2[i] + k3[i]) + (-gmm[i] + y[i])) / 6.0 + jm[i];
    }
    st_mat<D>(psi + (size_t)(t - 1) * DD, pk);
    st_vec<D>(lam + (size_t)(t - 1) * D, lk);
#pragma unroll
    for (int e = 0; e < DD; e++) { At[e] = Am[e]; Am[e] = An[e]; gst[e] = gsm[e]; gsm[e] = gsn[e]; }
#pragma unroll
    for (int i = 0; i < D; i++) { gmt[i] = gmm[i]; gmm[i] = gmn[i]; }
  }
}

template <int METHOD, bool FWD, int D>
hipError_t launch_d(const OdeArgs& a, hipStream_t st) {
  dim3 grid((a.batch + NTS - 1) / NTS), block(NTS);
  if (FWD) hipLaunchKernelGGL((k_fwd_small<METHOD, D>), grid, block, 0, st, a);
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

}  // namespace vgpa
