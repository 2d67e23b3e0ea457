// Generic (no symmetry assumption, any D <= 64, all four steppers) time-stepping kernels.
//
// One 256-thread workgroup integrates one problem over the whole grid; the D x D stage state, the
// stage's effective A and the stage output live in LDS, the step state S_k / Psi_t and the Runge-Kutta
// accumulators live in registers (each thread owns ceil(D*D/256) matrix entries).  This is the
// reference-faithful fallback (it evaluates A.X + X.A^T with both products exactly like
// src/numerics/ode_solver.py:60,94); the symmetric fast path is ode_mfma.hip.
//
// Reference semantics restated here:
//   forward : euler.py:27-92, heun.py:28-111, runge_kutta2.py:25-102 (incl. quirk Q2 at :96),
//             runge_kutta4.py:25-113
//   backward: euler.py:94-154, heun.py:113-190, runge_kutta2.py:104-194, runge_kutta4.py:115-211
//             (jumps added after the step, Q9; f_lam uses A.lam, Q3)
#include "vgpa_internal.h"

namespace vgpa {
namespace {

constexpr int NT = 256;

// effective A of a stage: either A0 or the mid-point 0.5*(A0 + A1) (same rounding as numpy)
template <bool MID>
__device__ __forceinline__ double a_eff(const double* __restrict__ a0, const double* __restrict__ a1, int idx) {
  if (MID) return 0.5 * (a0[idx] + a1[idx]);
  return a0[idx];
}

struct Lds {
  double* X;    // stage state           [D*D]
  double* Y;    // next stage state      [D*D]
  double* AB;   // stage effective A     [D*D]
  double* xv;   // stage vector          [D]
  double* yv;   // next stage vector     [D]
};

// Fill AB with the stage's effective A.  src may be a global pointer or (RK2 quirk) an LDS pointer.
template <bool MID>
__device__ __forceinline__ void fill_ab(double* AB, const double* a0, const double* a1, int DD) {
  for (int e = threadIdx.x; e < DD; e += NT) AB[e] = a_eff<MID>(a0, a1, e);
}

// rS[q] = f_S(X; AB) = -AB.X - X.AB^T + Sigma          (ode_solver.py:60)
template <int EPT>
__device__ __forceinline__ void rhs_fwd(const Lds& l, int D, const double (&sig)[EPT], double (&r)[EPT]) {
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = threadIdx.x + q * NT;
    double s1 = 0.0, s2 = 0.0;
    if (e < D * D) {
      const int i = e / D, j = e - i * D;
      for (int k = 0; k < D; k++) {
        s1 = __builtin_fma(l.AB[i * D + k], l.X[k * D + j], s1);
        s2 = __builtin_fma(l.X[i * D + k], l.AB[j * D + k], s2);
      }
    }
    r[q] = (-s1 - s2) + sig[q];
  }
}

// rS[q] = f_Psi(G, AB, X) = -G + X.AB + AB^T.X         (ode_solver.py:94)
template <int EPT>
__device__ __forceinline__ void rhs_bwd(const Lds& l, int D, const double (&g)[EPT], double (&r)[EPT]) {
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = threadIdx.x + q * NT;
    double s1 = 0.0, s2 = 0.0;
    if (e < D * D) {
      const int i = e / D, j = e - i * D;
      for (int k = 0; k < D; k++) {
        s1 = __builtin_fma(l.X[i * D + k], l.AB[k * D + j], s1);
        s2 = __builtin_fma(l.AB[k * D + i], l.X[k * D + j], s2);
      }
    }
    r[q] = (-g[q] + s1) + s2;
  }
}

// (AB . xv)[i] for thread i < D
__device__ __forceinline__ double matvec_row(const Lds& l, int D) {
  double s = 0.0;
  const int i = threadIdx.x;
  if (i < D)
    for (int k = 0; k < D; k++) s = __builtin_fma(l.AB[i * D + k], l.xv[k], s);
  return s;
}

template <int EPT>
__device__ __forceinline__ void store_stage(const Lds& l, int DD, const double (&v)[EPT]) {
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = threadIdx.x + q * NT;
    if (e < DD) l.Y[e] = v[q];
  }
}

__device__ __forceinline__ void swap_stage(Lds& l) {
  double* t = l.X; l.X = l.Y; l.Y = t;
  t = l.xv; l.xv = l.yv; l.yv = t;
}

// ------------------------------------------------------------------------------------------------
template <int METHOD, int EPT>
__global__ void __launch_bounds__(NT) k_fwd_generic(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int D = a.D, DD = D * D, Np = a.Np;
  const int prob = blockIdx.x;
  Lds l;
  l.X = smem; l.Y = smem + DD; l.AB = smem + 2 * DD; l.xv = smem + 3 * DD; l.yv = l.xv + D;
  const double* A = a.A + (size_t)prob * a.strideA;
  const double* b = a.b + (size_t)prob * a.strideB;
  double* mt = a.m + (size_t)prob * Np * D;
  double* st = a.S + (size_t)prob * Np * DD;
  const double dt = a.dt, h = 0.5 * a.dt;
  const int tid = threadIdx.x;
  const bool vth = tid < D;

  double sk[EPT], sig[EPT], r[EPT], acc1[EPT], acc2[EPT], tmp[EPT];
  double mk = 0.0;
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = tid + q * NT;
    sk[q] = (e < DD) ? a.S0[e] : 0.0;
    sig[q] = (e < DD) ? a.Sigma[e] : 0.0;
    if (e < DD) { st[e] = sk[q]; l.X[e] = sk[q]; }
  }
  if (vth) { mk = a.m0[tid]; mt[tid] = mk; l.xv[tid] = mk; }
  __syncthreads();

  for (int k = 0; k < Np - 1; k++) {
    const double* A0 = A + (size_t)k * DD;
    const double* A1 = A0 + DD;
    const double* b0 = b + (size_t)k * D;
    const double* b1 = b0 + D;
    double mnew = 0.0;
    if (METHOD == VGPA_ODE_EULER) {
      fill_ab<false>(l.AB, A0, A0, DD);
      __syncthreads();
      rhs_fwd<EPT>(l, D, sig, r);
      const double km = vth ? (-matvec_row(l, D) + b0[tid]) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) sk[q] = sk[q] + r[q] * dt;
      mnew = mk + km * dt;
    } else if (METHOD == VGPA_ODE_HEUN) {
      fill_ab<false>(l.AB, A0, A0, DD);
      __syncthreads();
      rhs_fwd<EPT>(l, D, sig, r);
      const double pm = vth ? (-matvec_row(l, D) + b0[tid]) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc1[q] = r[q]; tmp[q] = sk[q] + r[q] * dt; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = mk + pm * dt;
      __syncthreads();
      swap_stage(l);
      fill_ab<false>(l.AB, A1, A1, DD);
      __syncthreads();
      rhs_fwd<EPT>(l, D, sig, r);
      const double cm = vth ? (-matvec_row(l, D) + b1[tid]) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) sk[q] = sk[q] + h * (acc1[q] + r[q]);
      mnew = mk + h * (pm + cm);
    } else if (METHOD == VGPA_ODE_RK2) {
      // mean predictor uses A_k; covariance predictor uses S_k in the place of A_k (Q2)
      fill_ab<false>(l.AB, A0, A0, DD);
      __syncthreads();
      const double pm = vth ? (-matvec_row(l, D) + b0[tid]) : 0.0;
      __syncthreads();
      fill_ab<false>(l.AB, l.X, l.X, DD);
      __syncthreads();
      rhs_fwd<EPT>(l, D, sig, r);
#pragma unroll
      for (int q = 0; q < EPT; q++) tmp[q] = sk[q] + h * r[q];
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = mk + h * pm;
      __syncthreads();
      swap_stage(l);
      fill_ab<true>(l.AB, A0, A1, DD);
      __syncthreads();
      rhs_fwd<EPT>(l, D, sig, r);
      const double cm = vth ? (-matvec_row(l, D) + 0.5 * (b0[tid] + b1[tid])) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) sk[q] = sk[q] + dt * r[q];
      mnew = mk + dt * cm;
    } else {  // RK4
      // stage 1
      fill_ab<false>(l.AB, A0, A0, DD);
      __syncthreads();
      rhs_fwd<EPT>(l, D, sig, r);
      const double k1 = vth ? (-matvec_row(l, D) + b0[tid]) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc1[q] = r[q]; tmp[q] = sk[q] + h * r[q]; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = mk + h * k1;
      __syncthreads();
      swap_stage(l);
      // stage 2
      fill_ab<true>(l.AB, A0, A1, DD);
      __syncthreads();
      const double bmid = vth ? 0.5 * (b0[tid] + b1[tid]) : 0.0;
      rhs_fwd<EPT>(l, D, sig, r);
      const double k2 = vth ? (-matvec_row(l, D) + bmid) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc2[q] = r[q]; tmp[q] = sk[q] + h * r[q]; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = mk + h * k2;
      __syncthreads();
      swap_stage(l);
      // stage 3 (same effective A: AB is still the mid-point)
      rhs_fwd<EPT>(l, D, sig, r);
      const double k3 = vth ? (-matvec_row(l, D) + bmid) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc2[q] = acc2[q] + r[q]; tmp[q] = sk[q] + dt * r[q]; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = mk + dt * k3;
      __syncthreads();
      swap_stage(l);
      // stage 4
      fill_ab<false>(l.AB, A1, A1, DD);
      __syncthreads();
      rhs_fwd<EPT>(l, D, sig, r);
      const double k4 = vth ? (-matvec_row(l, D) + b1[tid]) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) sk[q] = sk[q] + dt * (acc1[q] + 2.0 * acc2[q] + r[q]) / 6.0;
      mnew = mk + dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0;
    }
    // publish step k+1
    mk = mnew;
    double* so = st + (size_t)(k + 1) * DD;
#pragma unroll
    for (int q = 0; q < EPT; q++) {
      const int e = tid + q * NT;
      if (e < DD) { so[e] = sk[q]; l.Y[e] = sk[q]; }
    }
    if (vth) { mt[(size_t)(k + 1) * D + tid] = mk; l.yv[tid] = mk; }
    __syncthreads();
    swap_stage(l);
  }
}

// jump (dE_obs) added after the step at index t-1
template <int EPT>
__device__ __forceinline__ void load_jump(const OdeArgs& a, int prob, int t1, int D, int DD, double (&js)[EPT],
                                          double& jm) {
  const int tid = threadIdx.x;
  if (a.js_dense) {
    const double* p = a.js_dense + ((size_t)prob * a.Np + t1) * DD;
#pragma unroll
    for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; js[q] = (e < DD) ? p[e] : 0.0; }
    jm = (tid < D) ? a.jm_dense[((size_t)prob * a.Np + t1) * D + tid] : 0.0;
  } else {
    const int n = a.obs_idx ? ldu(a.obs_idx, t1) : -1;           // (scalar load: see vgpa_internal.h)
#pragma unroll
    for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; js[q] = (n >= 0 && e < DD) ? a.js_const[e] : 0.0; }
    jm = (n >= 0 && tid < D) ? a.jm_sparse[((size_t)prob * a.n_obs + n) * D + tid] : 0.0;
  }
}

template <int METHOD, int EPT>
__global__ void __launch_bounds__(NT) k_bwd_generic(OdeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int D = a.D, DD = D * D, Np = a.Np;
  const int prob = blockIdx.x;
  Lds l;
  l.X = smem; l.Y = smem + DD; l.AB = smem + 2 * DD; l.xv = smem + 3 * DD; l.yv = l.xv + D;
  const double* A = a.A + (size_t)prob * a.strideA;
  const double* gm = a.dEm + (size_t)prob * Np * D;
  const double* gs = a.dEs + (size_t)prob * Np * DD;
  double* lam = a.lam + (size_t)prob * Np * D;
  double* psi = a.psi + (size_t)prob * Np * DD;
  const double dt = a.dt, h = 0.5 * a.dt;
  const int tid = threadIdx.x;
  const bool vth = tid < D;

  double pk[EPT], g[EPT], r[EPT], acc1[EPT], acc2[EPT], tmp[EPT], js[EPT];
  double lk = 0.0, jm = 0.0;
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    const int e = tid + q * NT;
    pk[q] = 0.0;
    if (e < DD) { psi[(size_t)(Np - 1) * DD + e] = 0.0; l.X[e] = 0.0; }
  }
  if (vth) { lam[(size_t)(Np - 1) * D + tid] = 0.0; l.xv[tid] = 0.0; }
  __syncthreads();

  for (int t = Np - 1; t > 0; t--) {
    const double* At = A + (size_t)t * DD;      // A_t
    const double* Am = At - DD;                 // A_{t-1}
    const double* gst = gs + (size_t)t * DD;
    const double* gsm = gst - DD;
    const double* gmt = gm + (size_t)t * D;
    const double* gmm = gmt - D;
    load_jump<EPT>(a, prob, t - 1, D, DD, js, jm);
    double lnew = 0.0;
    if (METHOD == VGPA_ODE_EULER) {
      fill_ab<false>(l.AB, At, At, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? gst[e] : 0.0; }
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double kl = vth ? (-gmt[tid] + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) pk[q] = pk[q] - r[q] * dt + js[q];
      lnew = lk - kl * dt + jm;
    } else if (METHOD == VGPA_ODE_HEUN) {
      fill_ab<false>(l.AB, At, At, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? gst[e] : 0.0; }
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double pl = vth ? (-gmt[tid] + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc1[q] = r[q]; tmp[q] = pk[q] - r[q] * dt; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = lk - pl * dt;
      __syncthreads();
      swap_stage(l);
      fill_ab<false>(l.AB, Am, Am, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? gsm[e] : 0.0; }
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double cl = vth ? (-gmm[tid] + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) pk[q] = pk[q] - h * (acc1[q] + r[q]) + js[q];
      lnew = lk - h * (pl + cl) + jm;
    } else if (METHOD == VGPA_ODE_RK2) {
      fill_ab<false>(l.AB, At, At, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? gst[e] : 0.0; }
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double pl = vth ? (-gmt[tid] + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) tmp[q] = pk[q] - h * r[q];
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = lk - h * pl;
      __syncthreads();
      swap_stage(l);
      fill_ab<true>(l.AB, Am, At, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? 0.5 * (gsm[e] + gst[e]) : 0.0; }
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double cl = vth ? (-(0.5 * (gmm[tid] + gmt[tid])) + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) pk[q] = pk[q] - dt * r[q] + js[q];
      lnew = lk - dt * cl + jm;
    } else {  // RK4
      fill_ab<false>(l.AB, At, At, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? gst[e] : 0.0; }
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double k1 = vth ? (-gmt[tid] + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc1[q] = r[q]; tmp[q] = pk[q] - h * r[q]; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = lk - h * k1;
      __syncthreads();
      swap_stage(l);
      // stages 2, 3 at the mid-point
      fill_ab<true>(l.AB, Am, At, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? 0.5 * (gsm[e] + gst[e]) : 0.0; }
      const double gmid = vth ? 0.5 * (gmm[tid] + gmt[tid]) : 0.0;
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double k2 = vth ? (-gmid + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc2[q] = r[q]; tmp[q] = pk[q] - h * r[q]; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = lk - h * k2;
      __syncthreads();
      swap_stage(l);
      rhs_bwd<EPT>(l, D, g, r);
      const double k3 = vth ? (-gmid + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) { acc2[q] = acc2[q] + r[q]; tmp[q] = pk[q] - dt * r[q]; }
      store_stage<EPT>(l, DD, tmp);
      if (vth) l.yv[tid] = lk - dt * k3;
      __syncthreads();
      swap_stage(l);
      // stage 4 at t-1
      fill_ab<false>(l.AB, Am, Am, DD);
#pragma unroll
      for (int q = 0; q < EPT; q++) { const int e = tid + q * NT; g[q] = (e < DD) ? gsm[e] : 0.0; }
      __syncthreads();
      rhs_bwd<EPT>(l, D, g, r);
      const double k4 = vth ? (-gmm[tid] + matvec_row(l, D)) : 0.0;
#pragma unroll
      for (int q = 0; q < EPT; q++) pk[q] = pk[q] - dt * (acc1[q] + 2.0 * acc2[q] + r[q]) / 6.0 + js[q];
      lnew = lk - dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0 + jm;
    }
    lk = lnew;
    double* po = psi + (size_t)(t - 1) * DD;
#pragma unroll
    for (int q = 0; q < EPT; q++) {
      const int e = tid + q * NT;
      if (e < DD) { po[e] = pk[q]; l.Y[e] = pk[q]; }
    }
    if (vth) { lam[(size_t)(t - 1) * D + tid] = lk; l.yv[tid] = lk; }
    __syncthreads();
    swap_stage(l);
  }
}

template <int METHOD, bool FWD>
hipError_t launch_m(const OdeArgs& a, hipStream_t st) {
  const int D = a.D, DD = D * D;
  const size_t lds = (size_t)(3 * DD + 2 * D) * sizeof(double);
  const int ept = (DD + NT - 1) / NT;
  dim3 grid(a.batch), block(NT);
#define VGPA_LAUNCH(E)                                                                       \
  do {                                                                                       \
    auto kern = FWD ? k_fwd_generic<METHOD, E> : k_bwd_generic<METHOD, E>;                    \
    if (lds > 48 * 1024)                                                                      \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                        \
  } while (0)
  if (ept <= 1) VGPA_LAUNCH(1);
  else if (ept <= 4) VGPA_LAUNCH(4);
  else if (ept <= 8) VGPA_LAUNCH(8);
  else VGPA_LAUNCH(16);
#undef VGPA_LAUNCH
  return hipGetLastError();
}

}  // namespace

hipError_t launch_ode_generic(int method, bool fwd, const OdeArgs& a, hipStream_t st) {
  if (a.D < 1 || a.D > kMaxSmallD) return hipErrorInvalidValue;
  switch (method) {
    case VGPA_ODE_EULER: return fwd ? launch_m<VGPA_ODE_EULER, true>(a, st) : launch_m<VGPA_ODE_EULER, false>(a, st);
    case VGPA_ODE_HEUN: return fwd ? launch_m<VGPA_ODE_HEUN, true>(a, st) : launch_m<VGPA_ODE_HEUN, false>(a, st);
    case VGPA_ODE_RK2: return fwd ? launch_m<VGPA_ODE_RK2, true>(a, st) : launch_m<VGPA_ODE_RK2, false>(a, st);
    case VGPA_ODE_RK4: return fwd ? launch_m<VGPA_ODE_RK4, true>(a, st) : launch_m<VGPA_ODE_RK4, false>(a, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace vgpa
