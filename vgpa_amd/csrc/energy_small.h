// E_sde integrand, <f>, dE_sde/dm, dE_sde/dS of ONE grid point for the models with closed-form Gaussian moments, as device
// functions on registers: used by the one-thread-per-grid-point kernels of energy.hip (operator-level API, vgpa_fetch) and by
// the fused lane-per-problem sweep of ode_small.hip, which evaluates them inside the backward recursion so that dEsde_dm,
// dEsde_dS, <f> and E_sde(t) never travel through HBM.
//
//   OU  : src/dynamics/ornstein_uhlenbeck.py:165-232
//   DW  : src/dynamics/double_well.py:169-260 with the Gaussian moments of src/var_bayes/gaussian_moments.py:43-183 (quirk Q7)
//   L63 : src/dynamics/lorenz_63.py:237-346, 348-568 (closed form, moments up to 4th order)
#pragma once
#include "vgpa_internal.h"

namespace vgpa {

struct Energy1d { double e_t, ef, edf, dm, ds, hyp; };

template <int MODEL>
__device__ __forceinline__ void energy_1d(double th, double sg, double la, double ob, double m, double s, Energy1d& r) {
  if (MODEL == VGPA_MODEL_OU) {
    const double ex2 = m * m + s;
    const double q1 = (th - la) * (th - la);
    r.e_t = ex2 * q1 + 2.0 * m * (th - la) * ob + (ob * ob);
    r.ef = -th * m;
    r.edf = -th;
    r.dm = (m * q1 + th * ob - la * ob) / sg;
    r.ds = 0.5 * q1 / sg;
    r.hyp = ex2 * (th - la) + m * ob;                    // ornstein_uhlenbeck.py:222-223
  } else {  // double well: f(x) = 4x(theta - x^2)
    const double c = 4.0 * th + la, c2 = c * c;
    const double m2 = m * m, m3 = m2 * m, m4 = m2 * m2, m5 = m4 * m, m6 = m3 * m3;
    const double s2 = s * s, s3 = s2 * s;
    const double ex2 = m2 + s;
    const double ex3 = m3 + 3 * m * s;
    const double ex4 = m4 + 6 * m2 * s + 3 * s2;
    const double ex6 = m6 + 15 * m4 * s + 45 * m2 * s2 + 15 * s3;
    r.e_t = 8.0 * (ex6 - c * ex4 + ob * ex3) + (c2 * ex2) - (2.0 * ob * c * m) + (ob * ob);
    r.ef = 4.0 * (th * m - ex3);
    r.edf = 4.0 * (th - 3.0 * ex2);
    const double dm2 = 2 * m, dm3 = 3 * (m2 + s), dm4 = 4 * (m3 + 3 * m * s);
    const double dm6 = 6 * (m5 + 10 * m3 * s + 15 * m * s2);
    const double ds2 = 1.0, ds3 = 3 * m, ds4 = 6 * (m2 + s), ds6 = 15 * m4 + 90 * m2 * s + 45 * s2;
    r.dm = 0.5 * (16.0 * dm6 - 8.0 * c * dm4 + 8.0 * ob * dm3 + c2 * dm2 - 2.0 * ob * c) / sg;
    r.ds = 0.5 * (16.0 * ds6 - 8.0 * c * ds4 + 8.0 * ob * ds3 + c2 * ds2) / sg;
    r.hyp = c * ex2 - 4.0 * ex4 - ob * m;                   // double_well.py:250-251
  }
}

// ds: the six distinct entries of the symmetric dE_sde/dS in the order xx, xy, xz, yy, yz, zz
struct EnergyL63 { double e_t, dm[3], ds[6], ef[3], hyp[6]; };

template <bool HYP>
__device__ __forceinline__ void energy_l63(const double* th, const double (&isg)[3], const double (&At)[9], const double (&bt)[3],
                                           const double (&mt)[3], const double (&St)[9], EnergyL63& r) {
#pragma clang fp contract(fast)
  const double vS = th[0], vR = th[1], vB = th[2];
  const double iSx = isg[0], iSy = isg[1], iSz = isg[2];
  const double A11 = At[0], A12 = At[1], A13 = At[2], A21 = At[3], A22 = At[4], A23 = At[5];
  const double A31 = At[6], A32 = At[7], A33 = At[8];
  const double b1 = bt[0], b2 = bt[1], b3 = bt[2];
  const double mx = mt[0], my = mt[1], mz = mt[2];
  const double Sxx = St[0], Sxy = St[1], Sxz = St[2], Syy = St[4], Syz = St[5], Szz = St[8];
  const double mx2 = mx * mx, my2 = my * my, mz2 = mz * mz;
  // 2nd order
  const double Exx = Sxx + mx2, Exy = Sxy + mx * my, Exz = Sxz + mx * mz;
  const double Eyy = Syy + my2, Eyz = Syz + my * mz, Ezz = Szz + mz2;
  // 3rd order
  const double Exxy = Sxx * my + 2 * Sxy * mx + mx2 * my;
  const double Exxz = Sxx * mz + 2 * Sxz * mx + mx2 * mz;
  const double Exyy = Syy * mx + 2 * Sxy * my + my2 * mx;
  const double Exzz = Szz * mx + 2 * Sxz * mz + mz2 * mx;
  const double Exyz = Sxy * mz + Sxz * my + Syz * mx + mx * my * mz;
  // 4th order
  const double Exxyy = Sxx * (my2 + Syy) + Syy * mx2 + 4.0 * Sxy * mx * my + (mx * my) * (mx * my) + 2 * (Sxy * Sxy);
  const double Exxzz = Sxx * (mz2 + Szz) + Szz * mx2 + 4.0 * Sxz * mx * mz + (mx * mz) * (mx * mz) + 2 * (Sxz * Sxz);
  const double vS2 = vS * vS, vR2 = vR * vR, vB2 = vB * vB;
  // <(f-g)^2> per component, lorenz_63.py:414-436
  const double EX = vS2 * (Eyy + Exx - 2 * Exy) + (A11 * A11) * Exx + (A12 * A12) * Eyy + (A13 * A13) * Ezz + b1 * b1 +
                    2 * (A11 * A12 * Exy + A11 * A13 * Exz - b1 * A11 * mx + A12 * A13 * Eyz - b1 * A12 * my -
                         b1 * A13 * mz +
                         vS * (A11 * Exy + A12 * Eyy + A13 * Eyz - b1 * my - A11 * Exx - A12 * Exy - A13 * Exz + b1 * mx));
  const double EY = vR2 * Exx + Eyy + Exxzz + (A21 * A21) * Exx + (A22 * A22) * Eyy + (A23 * A23) * Ezz + b2 * b2 +
                    2 * (Exyz - A21 * Exy - A22 * Eyy - A23 * Eyz - A21 * Exxz - A22 * Exyz - A23 * Exzz +
                         A21 * A22 * Exy + A21 * A23 * Exz + A22 * A23 * Eyz -
                         vR * (Exy + Exxz - A21 * Exx - A22 * Exy - A23 * Exz) -
                         b2 * (vR * mx - my - Exz + A21 * mx + A22 * my + A23 * mz));
  const double EZ = Exxyy + vB2 * Ezz + (A31 * A31) * Exx + (A32 * A32) * Eyy + (A33 * A33) * Ezz + b3 * b3 +
                    2 * (A31 * Exxy + A32 * Exyy + A33 * Exyz + A31 * A32 * Exy + A31 * A33 * Exz + A32 * A33 * Eyz -
                         vB * (Exyz + A31 * Exz + A32 * Eyz + A33 * Ezz) -
                         b3 * (Exy - vB * mz + A31 * mx + A32 * my + A33 * mz));
  r.e_t = 0.5 * (iSx * EX + iSy * EY + iSz * EZ);
  // d/dm (lorenz_63.py:497-531); d<xx>/dmx = 2mx, d<xy>/dmx = my, d<xz>/dmx = mz, ...
  const double dxx = 2.0 * mx, dyy = 2.0 * my, dzz = 2.0 * mz;
  const double dmx1 = dxx * (vS2 + A11 * A11) +
                      2 * (my * (-vS2 + vS * A11 - vS * A12 + A11 * A12) + mz * (A11 - vS) * A13 - vS * A11 * dxx +
                           b1 * (vS - A11));
  const double dmx2 = 2.0 * Exzz + dxx * (vR2 + A21 * A21) +
                      2 * (my * (-vR + vR * A22 - A21 + A21 * A22) + mz * (vR * A23 + b2 + A21 * A23) +
                           Eyz * (1 - A22) - vR * (2.0 * Exz) + vR * A21 * dxx - A21 * (2.0 * Exz) - A23 * Ezz -
                           b2 * (vR + A21));
  const double dmx3 = 2.0 * Exyy + (A31 * A31) * dxx +
                      2 * (my * (A31 * A32 - b3) + mz * (A33 - vB) * A31 + Eyz * (A33 - vB) + A31 * (2.0 * Exy) +
                           A32 * Eyy - A31 * b3);
  const double dmy1 = dyy * (vS2 + A12 * A12) +
                      2 * (mx * (-vS2 + vS * A11 - vS * A12 + A11 * A12) + mz * (vS + A12) * A13 + vS * A12 * dyy -
                           b1 * (vS + A12));
  const double dmy2 = dyy * (1 + A22 * A22) +
                      2 * (mx * (-vR + vR * A22 - A21 + A21 * A22) + Exz * (1 - A22) - A22 * dyy +
                           mz * (A22 * A23 - A23) + b2 * (1 - A22));
  const double dmy3 = 2.0 * Exxy + (A32 * A32) * dyy +
                      2 * (Exz * (A33 - vB) + A31 * Exx + A32 * (2.0 * Exy) + mx * (A31 * A32 - b3) +
                           mz * (A33 - vB) * A32 - A32 * b3);
  const double dmz1 = (A13 * A13) * dzz + 2 * (my * (vS + A12) + mx * (A11 - vS) - b1) * A13;
  const double dmz2 = 2.0 * Exxz + (A23 * A23) * dzz +
                      2 * (Exx * (-vR - A21) + mx * (vR * A23 + b2 + A21 * A23) + Exy * (1 - A22) +
                           my * (A22 * A23 - A23) - A23 * (2.0 * Exz + b2));
  const double dmz3 = dzz * (vB2 + A33 * A33) +
                      2 * ((A33 - vB) * (Exy + mx * A31 + my * A32 - b3) - vB * A33 * dzz);
  r.dm[0] = (0.5 * dmx1) * iSx + (0.5 * dmx2) * iSy + (0.5 * dmx3) * iSz;
  r.dm[1] = (0.5 * dmy1) * iSx + (0.5 * dmy2) * iSy + (0.5 * dmy3) * iSz;
  r.dm[2] = (0.5 * dmz1) * iSx + (0.5 * dmz2) * iSy + (0.5 * dmz3) * iSz;
  // d/dS (lorenz_63.py:535-566)
  const double dSxx = iSx * ((vS - A11) * (vS - A11)) + iSy * (Ezz + ((vR + A21) * (vR + A21)) - 2 * mz * (vR + A21)) +
                      iSz * (Eyy + (A31 * A31) + 2 * A31 * my);
  const double dSxy = iSx * 2 * (vS * A11 - vS2 - vS * A12 + A11 * A12) +
                      iSy * 2 * ((vR * A22 - vR - A21 + A21 * A22) + mz * (1 - A22)) +
                      iSz * (4.0 * Exy + 2 * (mz * (A33 - vB) + A31 * (2.0 * mx) + A32 * (2.0 * my) + (A31 * A32 - b3)));
  const double dSxz = iSx * 2 * (A11 - vS) * A13 +
                      iSy * (4.0 * Exz + 2 * ((vR * A23 + b2 + A21 * A23) + my * (1 - A22) - (2.0 * mx) * (vR + A21) -
                                              A23 * (2.0 * mz))) +
                      iSz * 2 * ((A33 - vB) * A31 + my * (A33 - vB));
  const double dSyy = iSx * ((vS + A12) * (vS + A12)) + iSy * ((1 - A22) * (1 - A22)) +
                      iSz * (Exx + (A32 * A32) + 2 * A32 * mx);
  const double dSyz = iSx * 2 * (vS + A12) * A13 + iSy * 2 * (mx * (1 - A22) + (A22 - 1) * A23) +
                      iSz * 2 * (mx * (A33 - vB) + (A33 - vB) * A32);
  const double dSzz = iSx * (A13 * A13) + iSy * (Exx + (A23 * A23) - 2 * A23 * mx) + iSz * ((vB - A33) * (vB - A33));
  r.ds[0] = 0.5 * dSxx; r.ds[1] = 0.5 * dSxy; r.ds[2] = 0.5 * dSxz;
  r.ds[3] = 0.5 * dSyy; r.ds[4] = 0.5 * dSyz; r.ds[5] = 0.5 * dSzz;
  // <f>, lorenz_63.py:319-321 (uses S[2,0] and S[1,0])
  r.ef[0] = vS * (my - mx);
  r.ef[1] = vR * mx - my - St[6] - mx * mz;
  r.ef[2] = St[3] + mx * my - vB * mz;
  if (HYP) {   // <(f-g)' df/dtheta> (lorenz_63.py:572-633) and <(f-g)^2> per component (:343)
    r.hyp[0] = Eyy * (vS + A12) + Exx * (vS - A11) + Exy * (A11 - 2 * vS - A12) + A13 * (Eyz - Exz) + b1 * (mx - my);
    r.hyp[1] = vR * Exx - Exy - Exxz + A21 * Exx + A22 * Exy + A23 * Exz - b2 * mx;
    r.hyp[2] = -Exyz + vB * Ezz - A31 * Exz - A32 * Eyz - A33 * Ezz + b3 * mz;
    r.hyp[3] = EX; r.hyp[4] = EY; r.hyp[5] = EZ;
  }
}

// Gradient of the Lagrangian at ONE grid point (variational.py:263-288), D <= 4, everything in registers:
//   dEb = Sigma^-1 (-<f> - A m + b),  Q = Sigma^-1 (<df/dx> + A) - 2 Psi,  gLb = dt (dEb + lam),  gLa = dt (Q S - (dEb + lam) m^T)
template <int D>
__device__ __forceinline__ void grad_point(const double (&Av)[D * D], const double (&bv)[D], const double (&mv)[D],
                                           const double (&Sv)[D * D], const double (&ef)[D], const double (&Ev)[D * D],
                                           const double (&Pv)[D * D], const double (&lm)[D], const double (&Iv)[D * D], double dt,
                                           double (&gA)[D * D], double (&gB)[D]) {
#pragma clang fp contract(fast)
  constexpr int DD = D * D;
  double rv[D], uv[D], pa[DD], q[DD];
#pragma unroll
  for (int i = 0; i < D; i++) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; k++) s = __builtin_fma(Av[i * D + k], mv[k], s);
    rv[i] = -ef[i] - s + bv[i];
  }
#pragma unroll
  for (int e = 0; e < DD; e++) pa[e] = Ev[e] + Av[e];
#pragma unroll
  for (int i = 0; i < D; i++) {
    double deb = 0.0;
#pragma unroll
    for (int l = 0; l < D; l++) deb = __builtin_fma(Iv[i * D + l], rv[l], deb);
    uv[i] = deb + lm[i];
#pragma unroll
    for (int j = 0; j < D; j++) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < D; l++) s = __builtin_fma(Iv[i * D + l], pa[l * D + j], s);
      q[i * D + j] = s - 2.0 * Pv[i * D + j];
    }
  }
#pragma unroll
  for (int i = 0; i < D; i++) {
    gB[i] = dt * uv[i];
#pragma unroll
    for (int j = 0; j < D; j++) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; k++) s = __builtin_fma(q[i * D + k], Sv[k * D + j], s);
      gA[i * D + j] = dt * (s - uv[i] * mv[j]);
    }
  }
}

}  // namespace vgpa
