// Internal declarations shared by the translation units of libvgpa_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vgpa_hip.h"

// ONE switch for every diagnostic or rejected-experiment code path of the library: -DVGPA_EXPERIMENTS.  vgpa_amd/build.py never sets it;
// a library built with it says so (bit 16 of vgpa_abi_version(), vgpa_amd/_lib.py refuses to load it unless VGPA_ALLOW_DIAGNOSTIC=1).
// Without it the wrong-result ablation macros and the cycle stamps of the micro-benchmarks (tools/ubench/) are compile errors, and the
// measured-slower kernel variants (eight product waves per problem, the outer-product cover, persistent energy waves) are not
// compiled at all -- their environment switches are then ignored.
#ifndef VGPA_EXPERIMENTS
#if defined(VGPA_ABL_NOFRAG) || defined(VGPA_ABL_NOSTORE) || defined(VGPA_ABL_NOVEC) || defined(VGPA_ABL_NOLOAD) || defined(VGPA_GF_ABL) || \
    defined(VGPA_STAMPS) || defined(VGPA_STAMPS_ROLE) || defined(VGPA_ENERGY_TPW) || defined(VGPA_ENERGY_TRACE)
#error "diagnostic / wrong-result switches need -DVGPA_EXPERIMENTS (the library then reports a diagnostic build)"
#endif
#endif

namespace vgpa {

#ifdef __HIPCC__
// A wave-uniform int from read-only global memory as a SCALAR load (s_load_dword -> SGPR, counted by lgkmcnt).  Behind the first
// global store of a kernel the compiler no longer proves such a load unclobbered and emits a vector load + s_waitcnt vmcnt(0) +
// v_readfirstlane on the spot -- i.e. it waits for every HBM prefetch the step has just issued (round 3: this was the backward
// steppers' observation-index lookup, one exposed memory round trip per time step).
__device__ __forceinline__ int ldu(const int32_t* p, int i) {
  typedef const int32_t __attribute__((address_space(4))) * cptr;
  return ((cptr)p)[i];
}
// ... and a wave-uniform double (constant matrices read once per kernel: they stay in SGPRs)
__device__ __forceinline__ double lds_const(const double* p, int i) {
  typedef const double __attribute__((address_space(4))) * cptr;
  return ((cptr)p)[i];
}
#endif

#ifdef __HIPCC__
// packed lower triangles (OdeArgs::s_packed): element (r, c), c <= r, at tri_off(r) + c; tri_row(e) = the row of flat index e
__host__ __device__ __forceinline__ int tri_off(int r) { return r * (r + 1) / 2; }
__device__ __forceinline__ int tri_row(int e) {                     // e < 2^20
  int i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
  i += (tri_off(i + 1) <= e) ? 1 : 0;
  i -= (tri_off(i) > e) ? 1 : 0;
  return i;
}
__host__ __device__ __forceinline__ int tri_idx(int r, int c) { return r >= c ? tri_off(r) + c : tri_off(c) + r; }
#endif

constexpr int kMaxSmallD = 64;   // single-workgroup (LDS resident) stepping kernels
constexpr int kMaxLaneD = 4;     // one-lane-per-problem stepping kernels (ode_small.hip)
constexpr int kMaxTheta = 4;

// Arguments of the time-stepping kernels (fwd: moments, bwd: Lagrange multipliers).
struct OdeArgs {
  int D, Np, batch;
  int sym_units;            // D <= 44: symmetric-unit kernels instead of the role-specialised ones (VGPA_FLAG_SYM_UNITS)
  size_t strideA, strideB;  // elements between consecutive problems in A / b (x layout: len_x for both)
  double dt;
  // forward
  const double* A;       // [B][Np][D][D]
  const double* b;       // [B][Np][D]
  const double* m0;      // [D]
  const double* S0;      // [D][D]
  const double* Sigma;   // [D][D]
  double* m;             // [B][Np][D]
  double* S;             // [B][Np][D][D]
  // backward
  const double* dEm;     // [B][Np][D]
  const double* dEs;     // [B][Np][D][D]
  const double* jm_dense;  // [B][Np][D]      or nullptr (then the sparse form is used)
  const double* js_dense;  // [B][Np][D][D]   or nullptr
  const int32_t* obs_idx;  // [Np] -> observation counter n, or -1
  const double* jm_sparse; // [B][M][D]
  const double* js_const;  // [D][D]
  int n_obs;
  double* lam;           // [B][Np][D]
  double* psi;           // [B][Np][D][D]
  // fused sweeps of the batched symmetric-unit kernels (sym::stores_q) with Sigma = sigma^2 I: q_on = 1 and q_scale = 1 / sigma^2, and
  // `psi` receives Q''_t = q_scale A_t - 2 Psi_t (what the gradient assembly needs of A_t and Psi_t) instead of Psi_t
  int q_on;
  double q_scale;
  // lane-per-problem kernels of the fused small-D sweep (ode_small.hip): the moments in a layout the context owns, TIME-major with the
  // problem fastest -- entry e (the packed lower triangle of S_t, then m_t: W = D (D + 1) / 2 + D entries) of grid point t of problem p at msT[(t * W + e) * bpad + p] -- so that a
  // wave's 64 lanes read / write 512 contiguous bytes per entry straight from / into registers; nullptr: the [B][Np] arrays m / S
  double* msT;
  int bpad;
  // fused batched sweeps of the fragment-cover kernels (33 <= D <= 40): S_t leaves the forward kernel as its packed LOWER triangle,
  // row-major -- element (r, c), c <= r, of grid point t of problem p at S[(p * Np + t) * D (D + 1) / 2 + r (r + 1) / 2 + c] -- which is
  // all the energy kernel factorises and half of what the gradient assembly streams (vgpa_fetch unpacks it on demand)
  int s_packed;
  int ds_packed;            // dEs holds packed lower triangles (EnergyArgs::ds_packed): symmetric-unit cover kernels, backward
  const double* jmT;     // sparse vector jumps in the same spirit: entry i of observation n of problem p at jmT[(n * D + i) * bpad + p]
  // backward kernel with the gradient assembly on its helper waves (sym::fuses_grad: fragment-cover kernels, RK4, Sigma = sigma^2 I,
  // Lorenz-96, packed S_t): grad_on = 1, q_on = 1; the kernel reads S (packed), m, b, Ef, Am of every grid point beside its own
  // streams and writes g = dt [gLa | gLb] (variational.py:263-288) -- Psi_t / Q''_t are NOT stored, lam_t is
  int grad_on;
  const double* Ef;      // [B][Np][D] <f>_t            (energy kernel)
  const double* Am;      // [B][Np][D] A_t m_t          (energy kernel)
  double* g;             // [B][strideA]: problem p's [Np][D][D] gLa, then [Np][D] gLb (the caller's x layout)
};

// Fused lane-per-problem pass of the models with closed-form moments (OU, double well, Lorenz-63; ode_small.hip::k_sweep_lane):
// E_sde terms of every grid point re-evaluated in registers, backward recursion, gradient assembly, F -- from (A, b, m, S) alone.
struct LaneSweepArgs {
  OdeArgs o;               // A, b (x layout), m, S, dt, batch, Np, D, sparse jumps (obs_idx, jm_sparse, js_const, n_obs)
  int model;
  int want_grad;           // 0: F only (no recursion); 1: F and the gradient
  double theta[kMaxTheta];
  double sigma1;           // 1-D models: sigma
  double isg[4];           // diagonal of Sigma^-1
  double isig[16];         // Sigma^-1 [D][D]
  double e0, pre, div;     // F = e0 + pre * trapz(E_sde(t)) / div + E_obs
  const double* eobs;      // [B]
  double* esde;            // [B]
  double* f;               // [B]
  double* g;               // [B][Np*D*D + Np*D] (want_grad)
};

struct EnergyArgs {
  int model, D, Np, batch;
  double dt;
  double theta[kMaxTheta];
  double sigma1;            // 1-D models: sigma
  const double* isg;        // [D] diagonal of Sigma^-1
  size_t strideA, strideB;  // elements between consecutive problems in A / b
  const double* A;          // [B][Np][D][D] (problem stride strideA)
  const double* b;          // [B][Np][D]    (problem stride strideB)
  const double* m;          // [B][Np][D]
  const double* S;          // [B][Np][D][D]
  double* e_t;              // [B][Np] integrand of E_sde
  double* Ef;               // [B][Np][D]
  double* Edf;              // [B][Np][D][D] or nullptr
  double* dEm;              // [B][Np][D]
  double* dEs;              // [B][Np][D][D]
  int s_packed;             // S holds packed lower triangles (OdeArgs::s_packed)
  int ds_upper;             // L96, D <= 64: write only the upper triangle of dEs (row <= col) -- the consumer is a symmetric-unit backward
                            // kernel, which reads nothing else (fused sweeps; VGPA_FETCH_DESDE_DS mirrors it on the way out)
  int ds_packed;            // ... as PACKED lower triangles (element (r, c), c <= r, of the symmetric matrix at tri_off(r) + c; matrix
                            // stride tri_off(D)): k_energy_l96_r with the cover kernels behind it (OdeArgs::ds_packed)
  double* hyp;              // [B][Np][H] per-grid-point integrands of dEsde/dtheta, dEsde/dSigma (nullptr: skipped)
  double* Am;               // [B][Np][D] A_t m_t, a by-product the gradient assembly reuses (L96 kernel; may be nullptr)
  int32_t* status;          // [B] device status word (bit0: S_t not positive definite)
};

struct ObsArgs {
  int D, Np, batch, n_obs, single;
  const int64_t* obs_t;     // [M]
  const double* obs_y;      // [M][D]
  const double* Q;          // [D][D]  H R^-1 H^T     (1-D: 1/r)
  const double* K;          // [D][D]  H^T R^-1 H^T   (1-D: H/r ... see obs kernel)
  const double* rinv_diag;  // [D]     diag(R^-1)
  double obs_const;         // M*(d*log(2pi) + logdet R)
  const double* m;          // [B][Np][D]
  const double* S;          // [B][Np][D][D]
  double* jm_sparse;        // [B][M][D]
  double* eobs;             // [B]
  int s_packed;             // S holds packed lower triangles (OdeArgs::s_packed)
  int diag;                 // Q and K are diagonal (diagonal R, H = I)
  double* part;             // [B][M] per-observation terms of the n-D energy (grid-parallel variant) or nullptr
};

struct GradArgs {
  int model, D, Np, batch, sigma_diag;
  int scalar_product;       // diagnostics: VALU inner products instead of the matrix-core product (VGPA_FLAG_FORCE_GENERIC)
  double dt;
  double theta[kMaxTheta];
  const double* isig;       // [D][D] Sigma^-1
  size_t strideA, strideB;  // elements between consecutive problems in A / b
  const double* A; const double* b;
  const double* m; const double* S;
  const double* lam; const double* psi;
  int s_packed;             // S holds packed lower triangles (OdeArgs::s_packed)
  int psi_is_q;             // `psi` holds Q''_t = Sigma^-1 A_t - 2 Psi_t (OdeArgs::q_on): A is not read
  const double* Ef;
  const double* Am;         // [B][Np][D] A_t m_t left by the energy kernel, or nullptr (then recomputed)
  const double* Edf;        // dense [B][Np][D][D] or nullptr (then recomputed from the model)
  double* g;                // [B][Np*D*D + Np*D]
};

struct ReduceArgs {
  int Np, batch;
  double dt, pre, div, e0;
  const double* e_t;        // [B][Np]
  const double* eobs;       // [B]
  double* esde;             // [B]
  double* f;                // [B]
};

// launchers (each returns hipGetLastError()) -----------------------------------------------------
hipError_t launch_ode_generic(int method, bool fwd, const OdeArgs& a, hipStream_t st);
hipError_t launch_ode_small(int method, bool fwd, const OdeArgs& a, hipStream_t st);     // D <= kMaxLaneD
bool sweep_lane_supported(int model, int D);
hipError_t launch_sweep_lane(int method, const LaneSweepArgs& a, hipStream_t st);
// msT -> the [B][Np] arrays: all grid points (vgpa_fetch, the separate kernels) or only what the observation terms read
hipError_t launch_ms_untranspose(int D, int Np, int batch, int bpad, const double* msT, double* m, double* S, hipStream_t st);
// observation terms (E_obs, sparse vector jumps) of the fused lane pass: one lane per problem, moments from msT, jumps to jmT
hipError_t launch_obs_lane(const ObsArgs& a, const double* msT, int bpad, double* jmT, hipStream_t st);
hipError_t launch_ode_wave(int method, bool fwd, const OdeArgs& a, hipStream_t st);      // 2 <= D <= kMaxLaneD, few problems
bool ode_mfma_supported(int method, bool fwd, int D);
hipError_t launch_ode_mfma(int method, bool fwd, const OdeArgs& a, hipStream_t st);
bool sym_stores_q(int method, int D);      // the backward kernel launch_ode_mfma picks for sym_units honours OdeArgs::q_on
bool sym_fuses_grad(int method, int D);    // ... and OdeArgs::grad_on (the gradient assembly on its helper waves)
// Psi_t = (diag(isg) A_t - Q''_t) / 2 in place (A: problem stride strideA, grid-point stride D*D)
// the strict lower triangle of [batch * Np] D x D matrices from their upper one, in place
hipError_t launch_mirror_upper(size_t n_mat, int D, double* m, hipStream_t st);
// n_mat packed lower triangles [D (D + 1) / 2] -> full symmetric D x D matrices
hipError_t launch_unpack_lower(size_t n_mat, int D, const double* packed, double* full, hipStream_t st);
hipError_t launch_psi_from_q(int batch, int Np, int D, size_t strideA, const double* A, const double* isg, double* psi_q, hipStream_t st);
hipError_t launch_energy(const EnergyArgs& a, hipStream_t st);
hipError_t launch_obs(const ObsArgs& a, hipStream_t st);   // uses the grid-parallel variant when a.part != nullptr
hipError_t launch_obs_dense(const ObsArgs& a, const double* js_const, double* jm_dense, double* js_dense,
                            hipStream_t st);
hipError_t launch_grad(const GradArgs& a, hipStream_t st);
hipError_t launch_reduce(const ReduceArgs& a, hipStream_t st);
// out[p][h] = trapezoid over t of e[p][t][h]
hipError_t launch_trapz_multi(const double* e, int Np, int H, int batch, double dt, double* out, hipStream_t st);
hipError_t launch_edf(const EnergyArgs& a, hipStream_t st);   // dense <df/dx> on request

// large-D (per-stage GEMM) single-rank drivers, large_d.hip
namespace ld {
// address ranges of the arrays of a batched context and their per-problem strides (ld_set_batch; see large_d.hip)
struct BatchMap {
  int nb = 1, n = 0;
  struct R { const double* lo; const double* hi; size_t stride; } r[16];
  void add(const double* base, size_t per_problem) {
    if (base && per_problem && n < 16) { r[n].lo = base; r[n].hi = base + (size_t)nb * per_problem; r[n].stride = per_problem; n++; }
  }
  size_t stride(const double* p) const {
    for (int i = 0; i < n; i++) if (p >= r[i].lo && p < r[i].hi) return r[i].stride;
    return 0;
  }
};
void ld_set_batch(const BatchMap* map_or_null);      // thread-local; nullptr = one problem
void ld_set_literal_products(bool on);               // thread-local: non-symmetric inputs, both products of the slope literally
size_t ld_workspace_doubles(int D);
hipError_t ld_solve_fwd(int method, double dt, int D, int Np, const double* A, const double* b, const double* m0,
                        const double* S0, const double* Sigma, double* m, double* S, double* ws, hipStream_t st);
hipError_t ld_solve_bwd(int method, double dt, int D, int Np, const double* A, const double* gm, const double* gs,
                        const double* jm, const double* js, double* lam, double* psi, double* ws, hipStream_t st,
                        const int32_t* obs_idx_host = nullptr);
// optional rocBLAS backend of the plain stage GEMM (lib_gemm.cpp); `use_library_gemm` is read by the single-rank drivers
bool library_gemm_available();
hipError_t library_gemm(bool transa, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                        hipStream_t st);
extern thread_local bool use_library_gemm;
hipError_t ld_bwd_step(int method, double dt, int D, const double* At, const double* Am, const double* Gt, const double* Gm,
                       const double* gt, const double* gmm, const double* Pt, const double* lt, double* Pn, double* ln,
                       const double* Jn, const double* jn, double* ws, hipStream_t st);
size_t lde_workspace_doubles(int D, int nb);
int lde_batch(int D, double budget_bytes);
hipError_t lde_energy(int D, int Np, double theta, const double* isg, const double* A, const double* b, const double* m,
                      const double* S, double* e_t, double* Ef, double* Edf, double* dEm, double* dEs, int32_t* status,
                      double* ws, int nbmax, hipStream_t st, double* hyp = nullptr, hipStream_t side = nullptr)   /* hyp: [Np][2 D] integrands of dEsde_dtheta | dEsde_dSigma, or nullptr */;
hipError_t lde_grad(int D, int Np, double dt, const double* isg, const double* A, const double* b, const double* m,
                    const double* S, const double* lam, const double* psi, const double* Ef, double* gA, double* gB,
                    double* ws, int nbmax, hipStream_t st, const double* isig_dense = nullptr);   // [D][D] Sigma^-1 when it is not diagonal
}  // namespace ld

// device vector algebra for the SCG driver (vecops.hip), segmented over the batch
int vec_blocks_per_seg(long long seglen, int nseg);
hipError_t vec_reduce(int mode, const double* a, const double* b, int nseg, long long seglen, double* scratch, hipStream_t st);
hipError_t vec_axpby(int nseg, long long seglen, const double* alpha_dev, const double* x, const double* beta_dev,
                     const double* y, double* out, hipStream_t st);

// tiny host-side dense helpers (row-major, fp64) ---------------------------------------------------
bool host_cholesky_lower(int n, const double* a, double* l);          // uses the lower triangle of a
void host_lower_inverse(int n, const double* l, double* linv);
bool host_spd_inverse(int n, const double* a, double* ainv, double* logdet);
void host_matmul(int n, const double* a, const double* b, double* c, bool ta, bool tb);

}  // namespace vgpa
