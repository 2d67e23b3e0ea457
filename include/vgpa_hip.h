/*
 * vgpa_hip.h -- C ABI of libvgpa_hip.so: the MI355X (gfx950) implementation of VGPA's
 * forward-backward variational smoothing sweep.
 *
 * The reference (vrettasm/VGPA) is pure Python and has no FFI; its plugin boundary is the
 * duck-typed Python surface listed below.  Every entry point of this header names the reference
 * interface it stands behind (paths relative to the reference repository root):
 *
 *   vgpa_solve_fwd      <- FwdOde.__call__ -> <stepper>.solve_fwd   src/var_bayes/fwd_ode.py:45-65,
 *                          src/numerics/{euler.py:27,heun.py:28,runge_kutta2.py:25,runge_kutta4.py:25}
 *   vgpa_solve_bwd      <- BwdOde.__call__ -> <stepper>.solve_bwd   src/var_bayes/bwd_ode.py:45-65,
 *                          src/numerics/{euler.py:94,heun.py:113,runge_kutta2.py:104,runge_kutta4.py:115}
 *   vgpa_energy         <- <model>.energy(A, b, m, S, obs_t)        src/dynamics/ornstein_uhlenbeck.py:165,
 *                          double_well.py:169, lorenz_63.py:237, lorenz_96.py:316
 *   vgpa_obs_energy     <- GaussianLikelihood.__call__ / .gradients src/var_bayes/gaussian_like.py:69-243
 *   vgpa_free_energy    <- VarGP.free_energy(x)                     src/var_bayes/variational.py:141-200
 *   vgpa_gradient       <- VarGP.gradient(x, eval_fun=False)        src/var_bayes/variational.py:202-289
 *   vgpa_sweep          <- VarGP.gradient(x, eval_fun=True)  (what SCG calls, src/numerics/optim_scg.py:167)
 *   vgpa_fetch          <- VarGP.arg_out                            src/var_bayes/variational.py:292
 *
 * Conventions
 *   - plain C, no C++/torch types; all floating point data is IEEE fp64, C-contiguous;
 *   - every function returns VGPA_OK (0) or a negative vgpa_status; no exception crosses the
 *     boundary; vgpa_last_error() gives the message of the last failure on that context;
 *   - host-pointer entry points copy in/out and are synchronous on return;
 *     the *_dev entry points take DEVICE pointers (hipMalloc'ed by the caller, same device) and
 *     enqueue on the context's stream; call vgpa_synchronize() before reading results;
 *   - buffers are caller-owned; the context owns its device workspace; a context is not
 *     thread-safe (one host thread per context, like the reference's single-threaded use);
 *   - there is NO CPU fallback: without a HIP device vgpa_create fails with VGPA_ERR_DEVICE.
 *
 * Batching: a context may hold `batch` independent problems that share the configuration
 * (observations, m0, S0, ...) but have different variational parameters x.  All per-problem
 * arrays are then laid out problem-major: x[batch][len_x], mt[batch][Np][D], F[batch], ...
 */
#ifndef VGPA_HIP_H
#define VGPA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGPA_ABI_VERSION 2
/* vgpa_abi_version() of a library built with -DVGPA_EXPERIMENTS (diagnostic switches and rejected experiments compiled in; never the
   product build) carries this bit beside the version */
#define VGPA_ABI_DIAGNOSTIC_BUILD 0x10000

/* ---- environment switches -------------------------------------------------------------------------------------------------------
   Every getenv() of the library, read ONCE per process (the first time the code path is reached).  None of them changes a result
   beyond rounding; all exist for same-box comparison runs and tests.  Nothing in vgpa_amd/ sets any of them.

   VGPA_ODE_KERNEL=pe|sym     33 <= D <= 44 / D <= 44: keep the role-specialised steppers (pe) / take the symmetric-unit ones (sym)
   VGPA_SYM_RUNS=1            33 <= D <= 40: the run layout of the symmetric-unit steppers instead of the fragment cover (then no
                              Q'' stream, no packed S_t, no fused gradient)
   VGPA_SYM_HELPERS=0|1       fragment-cover steppers: helper waves off / on at every batch size (default: up to one problem per CU)
   VGPA_FUSED_GRAD=0|1        backward RK4 fragment-cover kernel: never / always assemble the gradient on its third wave set
                              (default: from 64 problems per context on)
   VGPA_S_PACKED=0            fused batched sweeps: S_t as whole matrices between the kernels (default: packed lower triangles)
   VGPA_DS_PACKED=0           ... dEsde_dS as upper triangles of whole matrices (default: packed lower triangles)
   VGPA_SHARD_CHUNKS=<n>      row-sharded recursion: sub-blocks of the pipelined gather (default 4; 0 = the serial schedule);
                              per shard: vgpa_shard_set_option
   VGPA_STAGE_FUSED=<n>       D > 64, one GPU: the latency version of the one-kernel Runge-Kutta stage up to D = n (0: never; default 512, and
                              only while a launch has at most 64 tile pairs)
   VGPA_STAGE_WIDE=<n>        ... the throughput version up to D = n (0: never; default 2048, and not where D >= 384 is a multiple of 64); with both 0: GEMM + stage
                              kernel as in rounds 1-4.  Read per call (tests run all three at one size)
   VGPA_STAGE_FULL=1          D > 64: the stage kernel over whole tiles instead of symmetric tile pairs
   VGPA_GEMM_SCALAR_LOADS=1   D > 64: the 8-byte-load GEMM kernels also for full tiles
   VGPA_GEMM_PF=0             D > 64: the two-register-set loop of the stage products (default: four sets, no load under a branch)
   VGPA_GEMM_BM=32|64|128     D > 64: rows of the product's block tile (default: by the number of workgroups)
   VGPA_LDE_TWO_STREAMS=0     D > 64, energy terms: Cholesky + inverse of a batch on one stream (default: two half-batches on two streams)
   VGPA_LDE_INVERSE=rows      ... the inverse of L block row by block row (default: by halves, log2(D / 64) levels)
   VGPA_LDE_DIAG=valu         ... the 64 x 64 diagonal blocks on the vector ALU (default: matrix cores, k_diag64m)
   VGPA_LDE_TILE_MAP=0        ... workgroup -> tile of the batched products as launched (default: XCD-balanced maps)
   VGPA_LDE_SYRK_MIRROR=0     ... dEsde_dS from all its tiles (default: tiles on and below the diagonal, mirrored on the way out)
   VGPA_LDE_K_DOWN=0          ... products with a lower-triangular right operand: k loop upwards from each tile's own start (default: down)
   VGPA_LDE_GRAD_EPILOGUE=0   D > 64, gradient: the rank-one term and dt in a pass over the product Q S (default: in the product's epilogue)
   VGPA_LDE_PANEL=<blocks>    ... outer panel of the blocked Cholesky: one trailing update per <blocks> diagonal blocks (default 4; 1: one per block)
                              (the eight: A/B measurements of round 5, tools/trace_lde.sh; same results to 1e-9, the first and the fourth
                              bit for bit)
   VGPA_DIAG_REPEAT=<phase>:<n>  launch one phase (fwd|energy|bwd|grad) of the fused sweep n times (clock / power samples under one
                              kernel, tools/power_per_kernel.sh); every phase is a pure function of its inputs
   Only in builds with -DVGPA_EXPERIMENTS (vgpa_abi_version() carries VGPA_ABI_DIAGNOSTIC_BUILD; never the product build):
   VGPA_SYM_WAVES=8, VGPA_SYM_COVER=op (measured-and-rejected stepper variants); -DVGPA_LANE_T_EXPERIMENTS adds VGPA_LANE_T_FWD /
   VGPA_LANE_T_BWD (chunk lengths of the lane kernels, tools/lane_t_scan.sh).
   Host side: VGPA_LIB (vgpa_amd/_lib.py: path of another build of this library), VGPA_ALLOW_DIAGNOSTIC=1 (load a diagnostic build).  */

typedef enum {
  VGPA_OK = 0,
  VGPA_ERR_ARG = -1,        /* bad argument / inconsistent configuration (Python: ValueError)      */
  VGPA_ERR_DEVICE = -2,     /* no usable HIP device, or a HIP runtime call failed (RuntimeError)   */
  VGPA_ERR_NOT_PD = -3,     /* a matrix that must be positive definite is not (LinAlgError)        */
  VGPA_ERR_STATE = -4,      /* call order violated, e.g. gradient before free_energy (RuntimeError) */
  VGPA_ERR_UNSUPPORTED = -5, /* valid request that this build does not implement (NotImplementedError) */
  VGPA_ERR_COMM = -6        /* a collective of the row-sharded path failed or timed out; the communicator has been aborted
                               and the shard is unusable (RuntimeError on every rank)                */
} vgpa_status;

/* model ids: dynamical_systems registry, src/var_bayes/simulation.py:20-21 */
enum { VGPA_MODEL_NONE = -1, /* ODE-only context: solve_fwd / solve_bwd, no energy terms */
       VGPA_MODEL_OU = 0, VGPA_MODEL_DW = 1, VGPA_MODEL_L63 = 2, VGPA_MODEL_L96 = 3 };
/* stepper ids: num_integration registry, src/numerics/utilities.py:12-13 */
enum { VGPA_ODE_EULER = 0, VGPA_ODE_HEUN = 1, VGPA_ODE_RK2 = 2, VGPA_ODE_RK4 = 3 };
/* vgpa_fetch selectors: keys of VarGP.output, src/var_bayes/variational.py:189-196 */
enum {
  VGPA_FETCH_MT = 0, VGPA_FETCH_ST = 1, VGPA_FETCH_LAMT = 2, VGPA_FETCH_PSIT = 3,
  VGPA_FETCH_EFX = 4, VGPA_FETCH_EDF = 5, VGPA_FETCH_DESDE_DM = 6, VGPA_FETCH_DESDE_DS = 7,
  VGPA_FETCH_ESDE_T = 8 /* per-grid-point E_sde(t) before the trapezoid, (Np,) */
};
/* config flags */
enum {
  VGPA_FLAG_FORCE_GENERIC = 1, /* use the generic (no symmetry assumption) stepping kernels */
  VGPA_FLAG_LIBRARY_GEMM = 8,  /* D > 64: rocBLAS dgemm (dlopen'ed) for the plain stage products W = A.X / A^T.Psi instead of the
                                  hand-written MFMA GEMM; everything fused stays hand-written.  Off by default. */
  VGPA_FLAG_STREAM_LARGE_D = 4, /* D > 64: time-chunked sweep that keeps only x, S_t and the gradient resident (Psi_t and
                                  dEsde_dS_t live in chunk buffers; VGPA_FETCH_PSIT is unavailable).  Chosen automatically
                                  when the resident arrays would not fit into free device memory. */
  VGPA_FLAG_SYM_UNITS = 16,    /* 5 <= D <= 44: the symmetric-unit stepping kernels (ode_sym_impl.h; two problems per CU, the
                                  default for 44 < D <= 64) instead of the role-specialised ones.  Same results to rounding. */
  VGPA_FLAG_KEEP_PSI = 32,     /* batched symmetric-unit sweeps (33 <= D <= 40, RK2 / RK4, Sigma = sigma^2 I): by default the backward
                                  kernel leaves Q''_t = Sigma^-1 A_t - 2 Psi_t where Psi_t would be -- all the gradient assembly
                                  needs of the two, one HBM stream less -- and VGPA_FETCH_PSIT recovers Psi_t from it
                                  ((Sigma^-1 A_t - Q''_t) / 2, equal to rounding).  This flag stores Psi_t itself. */
  VGPA_FLAG_MATERIALIZE = 64   /* lane-per-problem contexts of OU / double well / Lorenz-63 (D = 1 always, D = 3 from 512 problems):
                                  by default the fused objective runs as TWO kernels -- forward moments, then one backward pass
                                  that re-evaluates the closed-form E_sde terms in registers, steps (lam, Psi), assembles the
                                  gradient and sums F -- and dEsde_dm / dEsde_dS / <f> / E_sde(t) / lam_t / Psi_t reach HBM only
                                  when vgpa_fetch asks for them.  This flag keeps the four-kernel path that writes them all. */
};

typedef struct vgpa_ctx vgpa_ctx;

typedef struct {
  int32_t abi_version;   /* VGPA_ABI_VERSION */
  int32_t device;        /* HIP device ordinal */
  int32_t model;         /* VGPA_MODEL_* */
  int32_t method;        /* VGPA_ODE_* */
  int32_t dim_d;         /* state dimension D (1 for OU/DW, 3 for L63, >=4 for L96) */
  int32_t n_pts;         /* grid points Np = len(arange(t0, tf+dt, dt)) */
  int32_t batch;         /* independent problems held by this context (>= 1) */
  int32_t flags;         /* VGPA_FLAG_* */
  double dt;             /* time step (> 0) */
  int32_t n_theta;       /* 1 (OU, DW, L96) or 3 (L63) */
  int32_t n_obs;         /* number of observation times M (may be 0 for ODE-only contexts) */
  const double* theta;   /* [n_theta] drift parameters                                      */
  const double* sigma;   /* [D*D] system noise covariance (1-D models: one value)           */
  const double* m0;      /* [D]   initial mean       (NULL for ODE-only contexts)           */
  const double* s0;      /* [D*D] initial covariance (NULL for ODE-only contexts)           */
  const int64_t* obs_t;  /* [M]   observation indices into the grid, strictly increasing    */
  const double* obs_y;   /* [M*D] observation values                                        */
  const double* obs_noise; /* [D*D] observation noise covariance R (1-D models: one value)  */
  const double* obs_h;   /* [D*D] observation operator H, or NULL for the identity          */
  double e0;             /* KL(q0||p0), constant in x (src/var_bayes/prior_kl0.py:46-92)    */
} vgpa_config;

/* What the library does NOT build (VGPA_ERR_UNSUPPORTED; the reference's numpy handles these at its own speed):
 *   - D > 64 for OU / DW / L63 (their D is 1 / 1 / 3 by definition) -- D > 64 exists for Lorenz-96 and for the bare ODE
 *     operators (model NONE);
 *   - batch > 1 in the TIME-CHUNKED large-D sweep (a batch that does not fit resident).  Everything else of D > 64 is built:
 *     dense Sigma / S0 / R / H, several problems per context (the per-stage kernels take them in grid.z: what fills the chip
 *     for 64 < D <= 512), the hyper-parameter members, non-symmetric operator-level inputs (both products of the slope
 *     formed literally; the fused sweep and the row-sharded drivers assume the symmetric S0 / Sigma every real run has);
 *   - VGPA_FETCH_PSIT / VGPA_FETCH_DESDE_DS in the time-chunked large-D sweep (they are never resident there).
 * The matrix-core stepping kernels cover D <= 64 with symmetric inputs; non-symmetric operator-level inputs run on the
 * generic LDS kernels (same results, ~15x slower at D = 40). */

/* lifetime ------------------------------------------------------------------------------- */
int vgpa_create(vgpa_ctx** out, const vgpa_config* cfg);
void vgpa_destroy(vgpa_ctx* ctx);
const char* vgpa_last_error(const vgpa_ctx* ctx);   /* ctx == NULL: last error of vgpa_create */
int vgpa_abi_version(void);
int vgpa_device_count(void);                         /* 0 when no HIP device is visible */
int vgpa_synchronize(vgpa_ctx* ctx);
void* vgpa_stream(vgpa_ctx* ctx);                    /* the context's hipStream_t */

/* operator level (host pointers; problem-major when batch > 1) ------------------------------ */
/* (m_t, S_t) from (A, b, m0, S0, Sigma).  A:[Np,D,D] b:[Np,D] m0:[D] s0:[D,D] sigma:[D,D] */
int vgpa_solve_fwd(vgpa_ctx* ctx, const double* lin_a, const double* off_b, const double* m0,
                   const double* s0, const double* sigma, double* mt, double* st);
/* (lam_t, Psi_t) from A, dEsde/dm [Np,D], dEsde/dS [Np,D,D] and the dense jump arrays of E_obs. */
int vgpa_solve_bwd(vgpa_ctx* ctx, const double* lin_a, const double* desde_dm, const double* desde_ds,
                   const double* deobs_dm, const double* deobs_ds, double* lam, double* psi);
/* E_sde and its per-grid-point terms.  Any output pointer may be NULL. */
int vgpa_energy(vgpa_ctx* ctx, const double* lin_a, const double* off_b, const double* mt,
                const double* st, double* esde, double* efx, double* edf, double* desde_dm,
                double* desde_ds);
/* E_obs and the dense jump arrays dEobs/dm [Np,D], dEobs/dS [Np,D,D] (zero off the obs rows). */
int vgpa_obs_energy(vgpa_ctx* ctx, const double* mt, const double* st, double* eobs,
                    double* deobs_dm, double* deobs_ds);

/* fused objective (state stays resident in HBM between calls) -------------------------------- */
int vgpa_free_energy(vgpa_ctx* ctx, const double* x, double* f);       /* f:[batch] */
int vgpa_gradient(vgpa_ctx* ctx, const double* x_or_null, double* g);  /* NULL: cached state */
int vgpa_sweep(vgpa_ctx* ctx, const double* x, double* f, double* g);  /* df(x, eval_fun=True) */
int vgpa_energy_parts(vgpa_ctx* ctx, double* e0, double* esde, double* eobs); /* each [batch] */
int vgpa_fetch(vgpa_ctx* ctx, int which, double* out);

/* device-pointer variants (x, g on the context's device; f written to HOST after a sync) ------ */
int vgpa_sweep_dev(vgpa_ctx* ctx, const double* x_dev, double* f_host, double* g_dev);
int vgpa_free_energy_dev(vgpa_ctx* ctx, const double* x_dev, double* f_host);
/* enqueue-only form for benchmarking / pipelining: F stays on the device (vgpa_fetch_f). */
int vgpa_sweep_enqueue(vgpa_ctx* ctx, const double* x_dev, double* g_dev);
int vgpa_fetch_f(vgpa_ctx* ctx, double* f_host);    /* syncs, checks the device status word */

/* vgpa_energy plus the two hyper-parameter members of <model>.energy()'s return tuple, which the reference computes but
 * nothing consumes (ornstein_uhlenbeck.py:222-226, double_well.py:250-254, lorenz_63.py:329-342, lorenz_96.py:421-434).
 * dEsde_dth: [B] (OU, DW), [B][3] (L63), [B][D] (L96); dEsde_dsig: [B] (1-D) or [B][D][D]; both NULL or both given
 * (D <= 64).  Every output may be NULL. */
int vgpa_energy_full(vgpa_ctx* ctx, const double* lin_a, const double* off_b, const double* mt, const double* st,
                     double* Esde, double* Efx, double* Edf, double* dEsde_dm, double* dEsde_ds,
                     double* dEsde_dth, double* dEsde_dsig);

/* gradient from the cached state into a DEVICE buffer (df(x) of SCG, src/numerics/optim_scg.py:100,235) */
int vgpa_gradient_dev(vgpa_ctx* ctx, double* g_dev);
/* LIFETIME of x_dev: the *_dev entry points consume the caller's x in place (zero copy) and the cached state keeps
 * referring to it -- vgpa_gradient_dev / vgpa_gradient(NULL) read A_t, b_t from that memory.  The caller keeps x_dev
 * alive and unchanged until the next evaluation, or calls vgpa_release_x before freeing / overwriting it: the cached
 * state is then dropped and a gradient request without x fails with VGPA_ERR_STATE instead of reading freed memory.
 * (vgpa_dev_free of the very pointer does the same by itself.) */
int vgpa_release_x(vgpa_ctx* ctx);

/* device-resident vector algebra for the SCG driver (src/numerics/optim_scg.py:75-285):
 * x, d and the gradients stay in HBM, only scalars return.  Every vector is the context's batch of `batch` segments of
 * `seglen` doubles (seglen = len(x) of one problem for SCG); results / coefficients are host arrays of `batch` doubles,
 * one per problem, so that `batch` independent optimisations advance in lock step.  Deterministic reductions. */
int vgpa_vec_dot(vgpa_ctx* ctx, const double* a_dev, const double* b_dev, uint64_t seglen, double* out_host);
int vgpa_vec_absmax(vgpa_ctx* ctx, const double* a_dev, uint64_t seglen, double* out_host);
int vgpa_vec_asum(vgpa_ctx* ctx, const double* a_dev, uint64_t seglen, double* out_host);
/* out = alpha[p]*x + beta[p]*y per problem p; y/beta may be NULL (out = alpha*x); out may alias x or y; a zero
 * coefficient drops its operand entirely (0*inf = 0), which is how finished problems are frozen */
int vgpa_vec_axpby(vgpa_ctx* ctx, uint64_t seglen, const double* alpha_host, const double* x_dev,
                   const double* beta_host_or_null, const double* y_dev_or_null, double* out_dev);

/* tuning knobs (tests / benchmarks); returns VGPA_ERR_ARG for an unknown option or a value out of range */
enum vgpa_option {
  VGPA_OPT_LD_CHUNK = 1        /* grid points per chunk of the time-chunked large-D sweep (>= 1) */
};
int vgpa_set_option(vgpa_ctx* ctx, int option, int64_t value);
/* E0 = KL(q0||p0): constant in x, but the reference recomputes it from the prior's current attributes on EVERY free_energy
 * call (src/var_bayes/variational.py:185; prior_kl0.py:30-92 reads self.mu0 / self.tau0) -- the host mirror hands the
 * current value over before each objective call instead of baking it into the context. */
int vgpa_set_prior_energy(vgpa_ctx* ctx, double e0);
/* 1 if the context runs the time-chunked large-D sweep (VGPA_FLAG_STREAM_LARGE_D or chosen for lack of memory) */
int vgpa_is_streaming(vgpa_ctx* ctx);

/* raw device memory helpers so that hosts without a HIP binding can own device buffers; whatever has not been returned
 * through vgpa_dev_free when the context is destroyed is freed with it */
int vgpa_dev_alloc(vgpa_ctx* ctx, uint64_t bytes, void** out);
int vgpa_dev_free(vgpa_ctx* ctx, void* ptr);
int vgpa_memcpy_h2d(vgpa_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int vgpa_memcpy_d2h(vgpa_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);

/* large-D (D > 64) stage-level entry points on DEVICE pointers --------------------------------------------------
 * One RK stage of the symmetric recursions = vgpa_ld_gemm (W[I_p,:] = A[I_p,:] X forward, (A^T)[I_p,:] Psi backward;
 * fp64 MFMA) + vgpa_ld_stage (fused element-wise update of the row block I_p = [row0, row0+Mp) and of the vector
 * recursion).  They stand behind the same reference code as vgpa_solve_fwd / vgpa_solve_bwd
 * (src/numerics/{euler,heun,runge_kutta2,runge_kutta4}.py); the host driver vgpa_amd/large_d.py places the RCCL
 * all-to-all / all-gather of the row-sharded recursion between the two calls.
 * `stream` is a hipStream_t (NULL = default stream).  C is written in column-chunk packed layout
 * C[(j / cw) * M * cw + i * cw + (j % cw)] (cw = N: plain row-major). */
typedef struct {
  int32_t D, row0, Mp, cw;      /* problem size, first row / number of rows of this rank, chunk width of W */
  int32_t fwd;                  /* 1: moments (S, m); 0: Lagrange multipliers (Psi, lam) */
  int32_t kstore;               /* 0 none, 1: K1 = R, 2: K23 = R, 3: K23 += R */
  int32_t final_mode;           /* 0: out = base +/- cx R; 1: base +/- cf R; 2: +/- cf (K1+R); 3: +/- cf (K1+2 K23+R)/6 */
  int32_t lda;                  /* leading dimension of A0 / A1 (vector recursion) */
  double cx, cf;
  const double* W;              /* [D/cw][Mp][cw] */
  const double* Wcol;           /* [D][Mp] (== W with one rank) */
  const double* E0;             /* [Mp][D] Sigma rows (fwd) or dEsde_dS[t] rows (bwd) */
  const double* E1;             /* NULL, or second operand of the mid-point 0.5*(E1+E0) */
  const double* J;              /* NULL, or [Mp][D] jump rows added by a final backward stage */
  const double* base;           /* [Mp][D] */
  double* K1; double* K23;      /* [Mp][D] */
  double* out;                  /* [Mp][D] */
  const double* A0;             /* full A of the stage (vector recursion reads rows row0..row0+Mp) */
  const double* A1;             /* NULL, or second operand of the mid-point */
  const double* x;              /* [D] stage vector */
  const double* e0; const double* e1;   /* [Mp] b (fwd) / dEsde_dm (bwd) entries; e1 NULL or mid-point operand */
  const double* jv;             /* NULL or [Mp] vector jump */
  const double* vbase;          /* [Mp] */
  double* k1v; double* k23v; double* vout;   /* [Mp] */
} vgpa_ld_stage_args;

int vgpa_ld_gemm(void* stream, int transa, int M, int N, int K, const double* A0, const double* A1_or_null, int lda,
                 const double* B, int ldb, double* C, int cw);
/* (vgpa_ld_stage runs the general kernel: E0 / E1 / J / base need not be symmetric.  The drivers inside the library, whose
 * inputs are checked to be symmetric, use a kernel that touches only the tiles on and above the diagonal.) */
int vgpa_ld_stage(void* stream, const vgpa_ld_stage_args* args);
/* One K-CHUNK launch of the stage product (the pipelined row-sharded stage splits a product into launches that wait for
 * different parts of the operand): C (+)= op(A)[:, ks] . B[ks, :] over the k-set ks = seg_tiles consecutive 16-wide k-tiles out
 * of every seg_stride k, K k's in all, starting where A0 / B point; accumulate != 0 continues the fp64 sums stored in C. */
int vgpa_ld_gemm_chunk(void* stream, int transa, int M, int N, int K, const double* A0, int lda, const double* B, int ldb,
                       double* C, int cw, int seg_tiles, int seg_stride, int accumulate);

/* row-sharded recursion for large D on the GPUs of one node (BASELINE configs[4]) ------------------
 * One vgpa_shard per process / GPU.  Rank p of `world` owns rows [p D/world, (p+1) D/world) of S_t / Psi_t inside every
 * Runge-Kutta stage and the contiguous slice [t_lo, t_hi) of the TIME grid of the results (vgpa_shard_time_slice): the
 * whole step / stage loop of src/numerics/{euler,heun,runge_kutta2,runge_kutta4}.py:solve_fwd / solve_bwd runs inside
 * vgpa_shard_solve_* with two collectives per stage and no host synchronisation: an all-to-all of the packed product
 * blocks on the compute stream and the gather of the next stage state's row blocks + vector entries -- either ONE grouped
 * all-gather on the compute stream (serial schedule) or, when the table has send / recv, C sub-blocks on a second stream
 * with the next stage's product split into C K-chunk launches that each wait for one sub-block only (pipelined schedule,
 * the default: VGPA_SHARD_OPT_GATHER_CHUNKS).  The collectives come through a vgpa_comm table: vgpa_rccl_comm_create fills
 * it from librccl (dlopen'ed; rank 0 calls vgpa_rccl_unique_id and the host runtime -- MPI, torch.distributed, a file --
 * hands the 128 bytes to the other ranks), tests inject their own.  All pointers are DEVICE pointers; the operator-level
 * calls take inputs replicated on every rank ([Np][D][D] / [Np][D], same meaning as vgpa_solve_fwd / vgpa_solve_bwd),
 * outputs hold only the rank's own grid points: m_own [t_hi - t_lo][D], S_own [t_hi - t_lo][D][D].  D must be a multiple
 * of `world`; symmetric S0 / Sigma / dEsde_dS / jumps as for every D > 64 path.  Calls return when the work is ENQUEUED
 * (vgpa_shard_synchronize waits, at most VGPA_SHARD_OPT_TIMEOUT_MS).
 * Errors are collective: a failing table entry, a launch failure or a time-out aborts the communicator (`abort`) and returns
 * VGPA_ERR_COMM / VGPA_ERR_DEVICE; the shard is unusable afterwards (every later call returns VGPA_ERR_COMM).  The host
 * runtime's restart policy is a fresh process or a non-zero exit -- never a re-exec of a process that holds the GPU. */
typedef struct vgpa_comm {
  void* user;
  /* recv[q * count .. (q+1) * count) = send of rank q; send may be recv + rank * count (in place) */
  int (*all_gather)(void* user, const double* send, double* recv, uint64_t count, void* stream);
  /* chunk q of send goes to rank q, chunk q of recv comes from rank q; count doubles per chunk */
  int (*all_to_all)(void* user, const double* send, double* recv, uint64_t count, void* stream);
  int (*group_begin)(void* user);      /* optional (may be NULL): fuse the calls up to group_end into one launch */
  int (*group_end)(void* user);
  /* optional point-to-point pair (both or neither; only between group_begin and group_end, which are then required): the
   * pipelined gather posts, per sub-block, one send to and one receive from every peer -- one xGMI link each */
  int (*send)(void* user, const double* buf, uint64_t count, int peer, void* stream);
  int (*recv)(void* user, double* buf, uint64_t count, int peer, void* stream);
  /* optional: tear the communicator down after a failure so that no rank stays inside a collective (ncclCommAbort) */
  int (*abort)(void* user);
} vgpa_comm;
typedef struct vgpa_shard vgpa_shard;
int vgpa_shard_create(vgpa_shard** out, int method, double dt, int dim_d, int n_pts, int rank, int world, int device,
                      const vgpa_comm* comm_or_null_if_world_1, void* stream_or_null);
void vgpa_shard_destroy(vgpa_shard* s);
int vgpa_shard_time_slice(const vgpa_shard* s, int* t_lo, int* t_hi);
/* the same rule without a shard: grid points [t_lo, t_hi) of `rank` out of `world` on a grid of n_pts (host arithmetic only) */
int vgpa_time_slice(int n_pts, int rank, int world, int* t_lo, int* t_hi);
void* vgpa_shard_stream(vgpa_shard* s);
int vgpa_shard_synchronize(vgpa_shard* s);
enum vgpa_shard_option {
  VGPA_SHARD_OPT_GATHER_CHUNKS = 1,  /* sub-blocks of the pipelined gather: 0 = serial schedule, 1..8 (reduced to what the
                                        row-block size allows: whole 16-row k-tiles per sub-block; 0 without send / recv).
                                        Default 4, or the environment's VGPA_SHARD_CHUNKS.  Same value on every rank. */
  VGPA_SHARD_OPT_TIMEOUT_MS = 2      /* bound of every host wait on the shard's streams (default 600000; <= 0: none) */
};
int vgpa_shard_set_option(vgpa_shard* s, int option, int64_t value);
int vgpa_shard_get_option(const vgpa_shard* s, int option, int64_t* value);
int vgpa_shard_solve_fwd(vgpa_shard* s, const double* lin_a, const double* off_b, const double* m0, const double* s0,
                         const double* sigma, double* m_own, double* s_own);
int vgpa_shard_solve_bwd(vgpa_shard* s, const double* lin_a, const double* desde_dm, const double* desde_ds,
                         const double* deobs_dm, const double* deobs_ds, double* lam_own, double* psi_own);
/* The fused sweep -- free energy AND gradient of VarGP (src/var_bayes/variational.py:141-288) -- of ONE Lorenz-96 problem
 * (diagonal system noise, diagonal R, H = I: the restrictions of every D > 64 path) on the row-sharded recursion:
 *   forward recursion, row-sharded, (m_t, S_t) time-sharded  ->  observation terms and E_sde terms of the rank's own grid
 *   points (time-parallel: lorenz_96.py:316-438 per grid point)  ->  a time -> row EXCHANGE of dEsde_dS (all-to-all: every
 *   rank receives rows I_p of every grid point, 1/world of an all-gather's bytes) + small all-gathers of dEsde_dm / E_sde(t)
 *   / the observation jumps  ->  backward recursion, row-sharded  ->  gradient of the own grid points.
 * vgpa_shard_sweep:         x_dev = [A_t (Np,D,D) | b_t (Np,D)] replicated on every rank (device).
 * vgpa_shard_sweep_sharded: x MEMORY-SHARDED like the gradient -- a_own [t_hi - t_lo][D][D], b_own [t_hi - t_lo][D] (the
 *                           layout of variational.py:153-162 restricted to the rank's grid points); rows I_p (forward) and
 *                           columns I_p (backward) of every A_t reach the recursions through two more exchanges.  No rank
 *                           ever holds a complete (Np, D, D) array: per rank seven arrays of Np/world matrices (x, the two
 *                           block copies of A, S, dEsde_dS / Psi, its row blocks, the gradient).
 * F comes back on every rank (host); the gradient stays TIME-sharded: grad_a_own [t_hi - t_lo][D][D], grad_b_own
 * [t_hi - t_lo][D] (device).  Synchronises the shard's streams before returning.  The return code is COLLECTIVE -- the same
 * on every rank: VGPA_ERR_NOT_PD when a marginal covariance S_t of ANY rank's grid points is not positive definite (the
 * reference raises LinAlgError, variational.py:380), VGPA_ERR_DEVICE when any rank could not allocate its buffers,
 * VGPA_ERR_ARG (before any collective) for observation indices that are out of range or not strictly increasing. */
typedef struct {
  double theta;                   /* Lorenz-96 forcing */
  const double* inv_sigma_diag;   /* [D] device: diagonal of Sigma^-1 */
  const double* m0;               /* [D] device */
  const double* s0;               /* [D][D] device */
  const double* sigma;            /* [D][D] device */
  int32_t n_obs;
  const int64_t* obs_t;           /* [n_obs] HOST: grid indices of the observations, strictly increasing */
  const double* obs_y;            /* [n_obs][D] device */
  const double* obs_rinv_diag;    /* [D] device: diagonal of R^-1 */
  double obs_const;               /* n_obs (D log(2 pi) + log det R)  (gaussian_like.py:87-92) */
  double e0;                      /* KL(q0||p0), constant in x */
} vgpa_shard_problem;
int vgpa_shard_sweep(vgpa_shard* s, const vgpa_shard_problem* problem, const double* x_dev, double* f_host,
                     double* grad_a_own, double* grad_b_own);
int vgpa_shard_sweep_sharded(vgpa_shard* s, const vgpa_shard_problem* problem, const double* a_own, const double* b_own,
                             double* f_host, double* grad_a_own, double* grad_b_own);
/* the two per-stage collectives alone (same buffers, sizes, streams and schedule as inside a stage), averaged over `reps`
 * rounds: what bench.py reports as per-stage collective milliseconds.  Collective call. */
int vgpa_shard_time_collectives(vgpa_shard* s, int reps, double* all_to_all_ms, double* gather_ms);
/* One stage of the forward RK4 recursion as the driver issues it (K-chunk products waiting for the previous gather's sub-blocks,
 * all-to-all, stage kernel, gather), `reps` times back to back on the shard's workspace; with_collectives == 0 leaves the
 * collectives out, so the difference of the two is the communication a stage does not hide.  Timing only (the workspace is
 * overwritten; call between sweeps).  Collective call when with_collectives != 0. */
int vgpa_shard_time_stage(vgpa_shard* s, int reps, int with_collectives, double* ms_per_stage);
/* Milliseconds of the six phases of this rank's LAST fused sweep (HIP events on the shard's stream): [0] exchanges of a
 * memory-sharded x, [1] forward recursion, [2] observation + E_sde terms of the own grid points, [3] small gathers + time -> row
 * exchange of dEsde_dS, [4] backward recursion, [5] gradient of the own grid points + F.  VGPA_ERR_STATE before the first sweep. */
int vgpa_shard_phase_ms(vgpa_shard* s, double* ms6);
/* RCCL behind the vgpa_comm table: collectives over xGMI; the library is dlopen'ed on first use */
#define VGPA_RCCL_UNIQUE_ID_BYTES 128
int vgpa_rccl_unique_id(void* out_128_bytes);
int vgpa_rccl_comm_create(vgpa_comm* out, const void* id_128_bytes, int rank, int world, int device);
void vgpa_rccl_comm_destroy(vgpa_comm* comm);
/* ranks of the communicator as librccl counts them (ncclCommCount); VGPA_ERR_ARG for a table RCCL did not fill */
int vgpa_rccl_comm_count(const vgpa_comm* comm, int* count);
/* communicators behind the table: 2 -- the collectives of the compute stream and the point-to-point groups of the communication stream
 * (the pipelined gather) each have their own (ncclCommSplit of the first) -- or 1 when the library cannot split */
int vgpa_rccl_comm_streams(const vgpa_comm* comm, int* count);

/* Device memory without a context -- what the callers of the row-sharded driver keep their operands and results in (the host
 * mirror vgpa_amd/large_d.py needs no tensor library for it).  kind: 1 = host -> device, 2 = device -> host, 3 = device -> device;
 * the copy is complete on return (it also waits for the device: results of an enqueued sweep are safe to read). */
int vgpa_device_alloc(int device, uint64_t bytes, void** out);
int vgpa_device_free(int device, void* ptr);
int vgpa_device_memcpy(int device, void* dst, const void* src, uint64_t bytes, int kind);

/* timing of the stepping kernel on the context's stream (HIP events), for bench.py's roofline */
int vgpa_profile_begin(vgpa_ctx* ctx);
int vgpa_profile_end(vgpa_ctx* ctx, double* fwd_ms, double* energy_ms, double* bwd_ms,
                     double* grad_ms, int64_t* n_sweeps);

#ifdef __cplusplus
}
#endif
#endif /* VGPA_HIP_H */
