// Time stepping for D = 2..4 with FEW problems per launch (Lorenz-63, one problem at a time: the reference's own use
// case): 16 lanes per problem, lane e < D*D owns matrix element (e / D, e % D), lanes e < D also own vector entry e;
// the products of a stage fetch their operands from the owning lanes with ds_bpermute -- no LDS memory, no barriers.
// Four problems share a wave.  Same expressions in the same order as ode_generic.hip (which restates the reference:
// euler.py, heun.py, runge_kutta2.py incl. quirk Q2, runge_kutta4.py; ode_solver.py:44-94), so the results equal the
// workgroup-per-problem kernels bit for bit; the step latency drops from ~2.8 k cycles (eight workgroup barriers per
// RK4 step) to the latency of the shuffles.  From 512 problems per launch ode_small.hip (one lane per problem) takes over.
// Everything a step reads from HBM is requested one step earlier.
#include "vgpa_internal.h"

namespace vgpa {
namespace {

constexpr int NTW = 64;    // one wave per workgroup = four problems of 16 lanes

// source lanes of a lane's operands: element (i, j) of the matrix, vector entry v
struct Src {
  int ik[4];   // (i, k)
  int kj[4];   // (k, j)
  int jk[4];   // (j, k)
  int ki[4];   // (k, i)
  int vk[4];   // (v, k)   row v of the matrix, for the mat-vec of the vector lanes
  int xk[4];   // lane that holds vector entry k
};

template <int D>
__device__ __forceinline__ Src make_src(int lane, int& e, bool& mat, bool& vec) {
  const int base = lane & ~15;
  e = lane & 15;
  mat = e < D * D;
  vec = e < D;
  const int i = mat ? e / D : 0, j = mat ? e % D : 0, v = vec ? e : 0;
  Src s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int kk = k < D ? k : 0;
    s.ik[k] = base + i * D + kk; s.kj[k] = base + kk * D + j; s.jk[k] = base + j * D + kk; s.ki[k] = base + kk * D + i;
    s.vk[k] = base + v * D + kk; s.xk[k] = base + kk;
  }
  return s;
}

__device__ __forceinline__ double sh(double v, int src) { return __shfl(v, src, 64); }

// f_S(X; AB)[i][j] = -AB.X - X.AB^T + Sigma
template <int D>
__device__ __forceinline__ double rhs_fwd(const Src& s, double ab, double x, double sig) {
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) {
    s1 = __builtin_fma(sh(ab, s.ik[k]), sh(x, s.kj[k]), s1);
    s2 = __builtin_fma(sh(x, s.ik[k]), sh(ab, s.jk[k]), s2);
  }
  return (-s1 - s2) + sig;
}

// f_Psi(G, AB, X)[i][j] = -G + X.AB + AB^T.X
template <int D>
__device__ __forceinline__ double rhs_bwd(const Src& s, double ab, double x, double g) {
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) {
    s1 = __builtin_fma(sh(x, s.ik[k]), sh(ab, s.kj[k]), s1);
    s2 = __builtin_fma(sh(ab, s.ki[k]), sh(x, s.kj[k]), s2);
  }
  return (-g + s1) + s2;
}

// (AB . xv)[v] on the vector lanes (every lane executes the shuffles)
template <int D>
__device__ __forceinline__ double matvec(const Src& s, double ab, double xv) {
  double y = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) y = __builtin_fma(sh(ab, s.vk[k]), sh(xv, s.xk[k]), y);
  return y;
}

// ------------------------------------------------------------------------------------------------
template <int METHOD, int D>
__global__ void __launch_bounds__(NTW) k_fwd_wave(OdeArgs a) {
  constexpr int DD = D * D;
  const int lane = threadIdx.x;
  int e; bool mat, vec;
  const Src s = make_src<D>(lane, e, mat, vec);
  const int pr = blockIdx.x * 4 + (lane >> 4);
  const bool live = pr < a.batch;
  const int prob = live ? pr : a.batch - 1;
  const int Np = a.Np;
  const int em = mat ? e : 0, ev = vec ? e : 0;
  const double* A = a.A + (size_t)prob * a.strideA + em;
  const double* b = a.b + (size_t)prob * a.strideB + ev;
  double* mt = a.m + (size_t)prob * Np * D + ev;
  double* st = a.S + (size_t)prob * Np * DD + em;
  const double dt = a.dt, h = 0.5 * a.dt;
  const bool wm = live && mat, wv = live && vec;

  double sk = a.S0[em], sig = a.Sigma[em], mk = a.m0[ev];
  if (wm) st[0] = sk;
  if (wv) mt[0] = mk;
  double a0 = A[0], b0 = b[0];
  double a1 = A[(Np > 1) ? DD : 0], b1 = b[(Np > 1) ? D : 0];

  for (int k = 0; k < Np - 1; k++) {
    const int kn = (k + 2 < Np) ? k + 2 : Np - 1;          // operands of the next step
    const double a2 = A[(size_t)kn * DD], b2 = b[(size_t)kn * D];
    double r, y;
    if (METHOD == VGPA_ODE_EULER) {
      r = rhs_fwd<D>(s, a0, sk, sig);
      y = matvec<D>(s, a0, mk);
      sk = sk + r * dt;
      mk = mk + (-y + b0) * dt;
    } else if (METHOD == VGPA_ODE_HEUN) {
      r = rhs_fwd<D>(s, a0, sk, sig);
      y = matvec<D>(s, a0, mk);
      const double pm = -y + b0, xv = mk + pm * dt, acc1 = r, x = sk + r * dt;
      r = rhs_fwd<D>(s, a1, x, sig);
      y = matvec<D>(s, a1, xv);
      sk = sk + h * (acc1 + r);
      mk = mk + h * (pm + (-y + b1));
    } else if (METHOD == VGPA_ODE_RK2) {
      y = matvec<D>(s, a0, mk);
      const double pm = -y + b0, xv = mk + h * pm;
      r = rhs_fwd<D>(s, sk, sk, sig);                        // S_k stands in for A_k (Q2)
      const double x = sk + h * r, am = 0.5 * (a0 + a1);
      r = rhs_fwd<D>(s, am, x, sig);
      y = matvec<D>(s, am, xv);
      sk = sk + dt * r;
      mk = mk + dt * (-y + 0.5 * (b0 + b1));
    } else {  // RK4
      const double bmid = 0.5 * (b0 + b1), am = 0.5 * (a0 + a1);
      r = rhs_fwd<D>(s, a0, sk, sig);
      y = matvec<D>(s, a0, mk);
      const double k1 = -y + b0, acc1 = r;
      double xv = mk + h * k1, x = sk + h * r;
      r = rhs_fwd<D>(s, am, x, sig);
      y = matvec<D>(s, am, xv);
      const double k2 = -y + bmid;
      double acc2 = r;
      xv = mk + h * k2; x = sk + h * r;
      r = rhs_fwd<D>(s, am, x, sig);
      y = matvec<D>(s, am, xv);
      const double k3 = -y + bmid;
      acc2 = acc2 + r;
      xv = mk + dt * k3; x = sk + dt * r;
      r = rhs_fwd<D>(s, a1, x, sig);
      y = matvec<D>(s, a1, xv);
      sk = sk + dt * (acc1 + 2.0 * acc2 + r) / 6.0;
      mk = mk + dt * (k1 + 2.0 * (k2 + k3) + (-y + b1)) / 6.0;
    }
    if (wm) st[(size_t)(k + 1) * DD] = sk;
    if (wv) mt[(size_t)(k + 1) * D] = mk;
    a0 = a1; a1 = a2; b0 = b1; b1 = b2;
  }
}

template <int METHOD, int D, bool DENSEJ>
__global__ void __launch_bounds__(NTW) k_bwd_wave(OdeArgs a) {
  constexpr int DD = D * D;
  const int lane = threadIdx.x;
  int e; bool mat, vec;
  const Src s = make_src<D>(lane, e, mat, vec);
  const int pr = blockIdx.x * 4 + (lane >> 4);
  const bool live = pr < a.batch;
  const int prob = live ? pr : a.batch - 1;
  const int Np = a.Np;
  const int em = mat ? e : 0, ev = vec ? e : 0;
  const double* A = a.A + (size_t)prob * a.strideA + em;
  const double* gm = a.dEm + (size_t)prob * Np * D + ev;
  const double* gs = a.dEs + (size_t)prob * Np * DD + em;
  double* lam = a.lam + (size_t)prob * Np * D + ev;
  double* psi = a.psi + (size_t)prob * Np * DD + em;
  const double dt = a.dt, h = 0.5 * a.dt;
  const bool wm = live && mat, wv = live && vec;

  double pk = 0.0, lk = 0.0;
  if (wm) psi[(size_t)(Np - 1) * DD] = 0.0;
  if (wv) lam[(size_t)(Np - 1) * D] = 0.0;
  // index t / t-1 of A, dEsde_dS, dEsde_dm (t-2 is requested inside the step)
  const int t1 = (Np > 1) ? Np - 2 : 0;
  double at = A[(size_t)(Np - 1) * DD], am1 = A[(size_t)t1 * DD];
  double gst = gs[(size_t)(Np - 1) * DD], gsm = gs[(size_t)t1 * DD];
  double gvt = gm[(size_t)(Np - 1) * D], gvm = gm[(size_t)t1 * D];
  // jumps of index t-1 (js, jm) and t-2 (requested inside the step); sparse: constant matrix jump at observation
  // indices, the index itself fetched two steps ahead through a VGPR (see ode_mfma_impl.h)
  const bool sparse = !DENSEJ && a.obs_idx;
  const double jsc = (!DENSEJ && a.js_const) ? a.js_const[em] : 0.0;
  int n_cur = (sparse && Np > 1) ? a.obs_idx[Np - 2] : -1;
  int n_next = (sparse && Np > 2) ? a.obs_idx[Np - 3] : -1;
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  int n_next2_v = -1;
  double js = 0.0, jm = 0.0;
  if (Np > 1) {
    if (DENSEJ) {
      js = a.js_dense[((size_t)prob * Np + (Np - 2)) * DD + em];
      jm = a.jm_dense[((size_t)prob * Np + (Np - 2)) * D + ev];
    } else if (n_cur >= 0) {
      jm = a.jm_sparse[((size_t)prob * a.n_obs + n_cur) * D + ev];
    }
  }

  for (int t = Np - 1; t > 0; t--) {
    if (t < Np - 1) n_next = __builtin_amdgcn_readfirstlane(n_next2_v);
    const int tn = (t >= 2) ? t - 2 : 0;
    const double an = A[(size_t)tn * DD], gsn = gs[(size_t)tn * DD], gvn = gm[(size_t)tn * D];
    n_next2_v = (sparse && t >= 3) ? a.obs_idx[t - 3 + vzero] : -1;
    double jsn = 0.0, jmn = 0.0;
    if (t >= 2) {
      if (DENSEJ) {
        jsn = a.js_dense[((size_t)prob * Np + tn) * DD + em];
        jmn = a.jm_dense[((size_t)prob * Np + tn) * D + ev];
      } else if (n_next >= 0) {
        jmn = a.jm_sparse[((size_t)prob * a.n_obs + n_next) * D + ev];
      }
    }
    if (!DENSEJ) js = (n_cur >= 0) ? jsc : 0.0;

    double r, y;
    if (METHOD == VGPA_ODE_EULER) {
      r = rhs_bwd<D>(s, at, pk, gst);
      y = matvec<D>(s, at, lk);
      pk = pk - r * dt + js;
      lk = lk - (-gvt + y) * dt + jm;
    } else if (METHOD == VGPA_ODE_HEUN) {
      r = rhs_bwd<D>(s, at, pk, gst);
      y = matvec<D>(s, at, lk);
      const double pl = -gvt + y, xv = lk - pl * dt, acc1 = r, x = pk - r * dt;
      r = rhs_bwd<D>(s, am1, x, gsm);
      y = matvec<D>(s, am1, xv);
      pk = pk - h * (acc1 + r) + js;
      lk = lk - h * (pl + (-gvm + y)) + jm;
    } else if (METHOD == VGPA_ODE_RK2) {
      r = rhs_bwd<D>(s, at, pk, gst);
      y = matvec<D>(s, at, lk);
      const double pl = -gvt + y, xv = lk - h * pl, x = pk - h * r;
      const double amid = 0.5 * (am1 + at), gmid = 0.5 * (gsm + gst);
      r = rhs_bwd<D>(s, amid, x, gmid);
      y = matvec<D>(s, amid, xv);
      pk = pk - dt * r + js;
      lk = lk - dt * (-(0.5 * (gvm + gvt)) + y) + jm;
    } else {  // RK4
      const double amid = 0.5 * (am1 + at), gmid = 0.5 * (gsm + gst), gvmid = 0.5 * (gvm + gvt);
      r = rhs_bwd<D>(s, at, pk, gst);
      y = matvec<D>(s, at, lk);
      const double k1 = -gvt + y, acc1 = r;
      double xv = lk - h * k1, x = pk - h * r;
      r = rhs_bwd<D>(s, amid, x, gmid);
      y = matvec<D>(s, amid, xv);
      const double k2 = -gvmid + y;
      double acc2 = r;
      xv = lk - h * k2; x = pk - h * r;
      r = rhs_bwd<D>(s, amid, x, gmid);
      y = matvec<D>(s, amid, xv);
      const double k3 = -gvmid + y;
      acc2 = acc2 + r;
      xv = lk - dt * k3; x = pk - dt * r;
      r = rhs_bwd<D>(s, am1, x, gsm);
      y = matvec<D>(s, am1, xv);
      pk = pk - dt * (acc1 + 2.0 * acc2 + r) / 6.0 + js;
      lk = lk - dt * (k1 + 2.0 * (k2 + k3) + (-gvm + y)) / 6.0 + jm;
    }
    if (wm) psi[(size_t)(t - 1) * DD] = pk;
    if (wv) lam[(size_t)(t - 1) * D] = lk;
    at = am1; am1 = an; gst = gsm; gsm = gsn; gvt = gvm; gvm = gvn;
    if (DENSEJ) js = jsn;
    jm = jmn; n_cur = n_next;
  }
}

template <int METHOD, bool FWD, int D>
hipError_t launch_d(const OdeArgs& a, hipStream_t st) {
  dim3 grid((a.batch + 3) / 4), block(NTW);
  if (FWD) hipLaunchKernelGGL((k_fwd_wave<METHOD, D>), grid, block, 0, st, a);
  else if (a.js_dense) hipLaunchKernelGGL((k_bwd_wave<METHOD, D, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_bwd_wave<METHOD, D, false>), grid, block, 0, st, a);
  return hipGetLastError();
}

template <int METHOD, bool FWD>
hipError_t launch_m(const OdeArgs& a, hipStream_t st) {
  switch (a.D) {
    case 2: return launch_d<METHOD, FWD, 2>(a, st);
    case 3: return launch_d<METHOD, FWD, 3>(a, st);
    case 4: return launch_d<METHOD, FWD, 4>(a, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_ode_wave(int method, bool fwd, const OdeArgs& a, hipStream_t st) {
  if (a.D < 2 || a.D > kMaxLaneD) return hipErrorInvalidValue;
  switch (method) {
    case VGPA_ODE_EULER: return fwd ? launch_m<VGPA_ODE_EULER, true>(a, st) : launch_m<VGPA_ODE_EULER, false>(a, st);
    case VGPA_ODE_HEUN: return fwd ? launch_m<VGPA_ODE_HEUN, true>(a, st) : launch_m<VGPA_ODE_HEUN, false>(a, st);
    case VGPA_ODE_RK2: return fwd ? launch_m<VGPA_ODE_RK2, true>(a, st) : launch_m<VGPA_ODE_RK2, false>(a, st);
    case VGPA_ODE_RK4: return fwd ? launch_m<VGPA_ODE_RK4, true>(a, st) : launch_m<VGPA_ODE_RK4, false>(a, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace vgpa
