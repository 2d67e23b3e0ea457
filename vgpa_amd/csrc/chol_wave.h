// One-wave Cholesky building blocks shared by the Lorenz-96 energy kernels of D <= 64 (energy.hip) and the 64 x 64 diagonal-block
// kernel of the blocked factorisation above D = 64 (large_d_energy.hip::k_diag64): the reciprocal square root without the library's
// class test, and the four pivots of a four-column panel with lane = row (src/numerics/utilities.py:239-310 chol_inv /
// numpy.linalg.cholesky on the lower triangle).
#pragma once
#include <hip/hip_runtime.h>

namespace vgpa {
namespace cholw {

// 1/sqrt(x) for x > 0: the operation sequence of the device library's rsqrt (v_rsq_f64 and one corrected Newton step, ~1 ulp)
// without its class test for 0 / inf -- the callers flag non-positive pivots themselves
__device__ __forceinline__ double rsqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(y * -x, y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
}

// The four pivots of Cholesky panel j0 .. j0+3 (lane = row; the panel already holds S - L[:, :j0] L[j0:j0+4, :j0]^T).  Every lane
// factors the 4x4 diagonal block itself from ten broadcast LDS reads and then solves its own row against it: the same operations in
// the same order as the lane-j-broadcasts-its-pivot form of rounds 1-2 (bit-identical factor), but the 40-pivot chain has no
// cross-lane step (v_readlane -> SGPR -> VALU hazards) and one predicated region per panel instead of eight.  myrd collects 1 / L[l][l]:
// a pivot that is not positive and finite leaves inf / NaN there (and NaN in the rest of the factor), which the caller tests once
// after the loop.
template <int LD>
__device__ __forceinline__ void chol_panel_pivots(double* Lm, double& myrd, int j0, int l, int li, bool pad) {
  const double* dg = Lm + j0 * LD + j0;
  const double* rowi = Lm + li * LD + j0;
  // Statement order = issue order, pinned by scheduling barriers: the block's columns and the row's entries are read one step ahead
  // of their use instead of all at the top (the kernel runs at its register limit, 168 for three waves per SIMD; a spill there is a
  // memory round trip per use).
  const double a00 = dg[0], a10 = dg[LD], a20 = dg[2 * LD], a30 = dg[3 * LD];
  const double a11 = dg[LD + 1], a21 = dg[2 * LD + 1], a31 = dg[3 * LD + 1];
  __builtin_amdgcn_sched_barrier(0);
  const double r0 = rsqrt_pos(a00);
  const double l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
  const double a22 = dg[2 * LD + 2], a32 = dg[3 * LD + 2], s0 = rowi[0], s1 = rowi[1];
  __builtin_amdgcn_sched_barrier(0);
  const double p1 = __builtin_fma(-l10, l10, a11);
  const double r1 = rsqrt_pos(p1);
  const double l21 = __builtin_fma(-l20, l10, a21) * r1, l31 = __builtin_fma(-l30, l10, a31) * r1;
  const double x0 = s0 * r0;
  const double x1 = __builtin_fma(-x0, l10, s1) * r1;
  myrd = (l == j0) ? r0 : ((l == j0 + 1) ? r1 : myrd);
  const double a33 = dg[3 * LD + 3], s2 = rowi[2], s3 = rowi[3];
  __builtin_amdgcn_sched_barrier(0);
  const double p2 = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, a22));
  const double r2 = rsqrt_pos(p2);
  const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, a32)) * r2;
  const double x2 = __builtin_fma(-x1, l21, __builtin_fma(-x0, l20, s2)) * r2;
  const double p3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, a33)));
  const double r3 = rsqrt_pos(p3);
  const double x3 = __builtin_fma(-x2, l32, __builtin_fma(-x1, l31, __builtin_fma(-x0, l30, s3))) * r3;
  myrd = (l == j0 + 2) ? r2 : ((l == j0 + 3) ? r3 : myrd);
  if (pad) {                                   // rows above the pivot: the strict upper triangle is zeroed on the way
    double* wr = Lm + l * LD + j0;
    wr[0] = (l >= j0) ? x0 : 0.0;
    wr[1] = (l >= j0 + 1) ? x1 : 0.0;
    wr[2] = (l >= j0 + 2) ? x2 : 0.0;
    wr[3] = (l >= j0 + 3) ? x3 : 0.0;
  }
}

}  // namespace cholw
}  // namespace vgpa
