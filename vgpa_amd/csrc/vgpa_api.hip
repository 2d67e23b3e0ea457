// C ABI of libvgpa_hip.so (see include/vgpa_hip.h for the contract and the reference interfaces).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vgpa_internal.h"

using namespace vgpa;

namespace {
thread_local std::string g_create_error;
}

struct vgpa_ctx {
  vgpa_config cfg{};
  int D = 0, Np = 0, B = 1, M = 0;
  size_t DD = 0, len_x = 0;
  bool single = false, full = false, sigma_diag = true, sym_inputs = true;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;      // side stream of the D > 64 energy terms (lde_energy's look-ahead), created on first use
  std::string err;
  double theta[kMaxTheta] = {0, 0, 0, 0};
  // device buffers
  std::vector<void*> allocs;
  const double* xcur = nullptr;   // where the kernels read [A|b] from (d_x or the caller's device buffer)
  double *d_x = nullptr, *d_m = nullptr, *d_S = nullptr, *d_Ef = nullptr, *d_dEm = nullptr, *d_dEs = nullptr, *d_Am = nullptr;
  double *d_lam = nullptr, *d_psi = nullptr, *d_g = nullptr, *d_et = nullptr, *d_eobs = nullptr, *d_esde = nullptr;
  double *d_f = nullptr, *d_jm = nullptr, *d_Edf = nullptr, *d_jm_dense = nullptr, *d_js_dense = nullptr;
  double *d_m0 = nullptr, *d_S0 = nullptr, *d_Sigma = nullptr, *d_isig = nullptr, *d_isg = nullptr;
  double *d_obs_y = nullptr, *d_Q = nullptr, *d_K = nullptr, *d_rinv = nullptr, *d_jsc = nullptr;
  double *d_op_m0 = nullptr, *d_op_S0 = nullptr, *d_op_Sigma = nullptr;
  double* d_ld_ws = nullptr;      // workspace of the large-D drivers
  double* d_lde_ws = nullptr;     // workspace of the large-D energy / gradient kernels
  // time-chunked ("streamed") large-D sweep: Psi_t and dEsde_dS_t live only in chunk buffers of ld_chunk + 1 matrices
  bool stream_ld = false;
  int ld_chunk = 0;
  double *d_dEs_c = nullptr, *d_psi_c = nullptr;
  bool psi_is_q = false;       // d_psi holds Q''_t = A_t / sigma^2 - 2 Psi_t (fused batched sweeps, OdeArgs::q_on)
  bool des_upper = false;      // d_dEs holds the upper triangles only (EnergyArgs::ds_upper)
  void* h_fs = nullptr;        // pinned host block [B doubles | B status words]: F and the status words come back in one round trip (vgpa_fetch_f)
  bool des_packed = false;     // d_dEs holds packed lower triangles (EnergyArgs::ds_packed); d_jscp = the constant matrix jump in the same layout
  double* d_jscp = nullptr;
  bool isg_iso = false;        // Sigma = sigma^2 I
  double isg0 = 1.0;           // 1 / sigma^2 then
  std::vector<int32_t> h_obs_idx; // host copy of obs_idx [Np]
  bool obs_diag = false;          // diagonal R and H = I: Q, K are diagonal
  double* d_obs_part = nullptr;   // [B][M] per-observation energy terms (large-D observation kernel)
  double* d_hyp = nullptr;        // [B][Np][H] integrands of the hyper-parameter gradients (vgpa_energy_hyper only)
  double* d_hypT = nullptr;       // [B][H] their trapezoids
  bool hyp_on = false;
  std::vector<double> h_isig;     // host copy of Sigma^-1 [D][D]
  double* d_vec_scratch = nullptr; // [2B coefficients | B results | B*bps partials] of the vector algebra
  size_t vec_scratch_n = 0;
  // axpby coefficients travel through a pinned ring (the caller's arrays may be temporaries that die on return)
  static constexpr int kCoefSlots = 16;
  double* h_coef = nullptr;        // pinned [kCoefSlots][2B]
  hipEvent_t ev_coef[kCoefSlots] = {};
  unsigned coef_next = 0;
  std::vector<void*> user_allocs;  // vgpa_dev_alloc memory not yet returned through vgpa_dev_free
  int lde_nb = 1;
  double lde_budget = 1.0e9;      // bytes the batched large-D energy workspace may take
  int64_t* d_obs_t = nullptr;
  int32_t *d_obs_idx = nullptr, *d_status = nullptr;
  double obs_const = 0.0, sigma1 = 1.0;
  bool have_state = false;
  bool bwd_stored = true;        // d_lam / d_psi hold the backward recursion of the cached state (false: F-only evaluation of a context whose backward
                                 // kernel assembles the gradient -- grad_fused_ok -- or that kernel, which keeps Psi_t to itself; vgpa_fetch materialises)
  bool derived_valid = true;     // dEsde_dm / dEsde_dS / <f> / E_sde(t) / lam / Psi belong to the cached (m, S): false behind a fused lane pass
  double* d_msT = nullptr;       // fused lane pass: the moments time-major, problem fastest (OdeArgs::msT), [Np][D(D+1)/2 + D][bpad]: packed lower triangle of S_t, then m_t
  double* d_jmT = nullptr;       // ... and its sparse vector jumps, [M][D][bpad]
  int bpad = 0;
  bool s_packed = false;         // d_S holds packed lower triangles (OdeArgs::s_packed): the fused batched sweeps of the cover kernels
  double* d_Sfull = nullptr;     // ... and their unpacked copy, made when vgpa_fetch (or a kernel that wants S_t whole) asks
  bool ms_valid = true;          // d_m / d_S hold the cached moments (false: only d_msT does; untransposed on demand)
  bool sym_units = false;        // stepping-kernel family of this context (pick_kernel_family)
  // profiling
  bool prof = false;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double prof_ms[4] = {0, 0, 0, 0};
  int64_t prof_n = 0;
  bool prof_pending = false;
};

namespace {

int fail(vgpa_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(c, expr)                                                                            \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail((c), VGPA_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
int dev_alloc(vgpa_ctx* c, T** p, size_t count) {
  void* q = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
  c->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return VGPA_OK;
}

// buffers that only some entry points need are allocated on first use
template <typename T>
int ensure(vgpa_ctx* c, T** p, size_t count) { return *p ? VGPA_OK : dev_alloc(c, p, count); }

template <typename T>
int upload(vgpa_ctx* c, T* dst, const T* src, size_t count) {
  HIP_TRY(c, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return VGPA_OK;
}

template <typename T>
int download(vgpa_ctx* c, T* dst, const T* src, size_t count) {
  HIP_TRY(c, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  return VGPA_OK;
}

bool is_symmetric(const double* a, int n) {
  double mx = 0.0, df = 0.0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      mx = std::fmax(mx, std::fabs(a[i * n + j]));
      df = std::fmax(df, std::fabs(a[i * n + j] - a[j * n + i]));
    }
  return df <= 1e-14 * mx;
}

bool stack_symmetric(const double* a, size_t count, int n) {
  for (size_t t = 0; t < count; t++)
    if (!is_symmetric(a + t * n * n, n)) return false;
  return true;
}

}  // namespace

// The variational parameters are consumed in the caller's x layout: problem p at xcur + p*len_x, A first, then b.
// xcur is either the context's own copy d_x (host entry points) or the caller's device buffer (zero copy).
static inline const double* ctx_A(vgpa_ctx* c) { return c->xcur; }
static inline const double* ctx_b(vgpa_ctx* c) { return c->xcur + (size_t)c->Np * c->DD; }

static int ingest_x(vgpa_ctx* c, const double* x, bool on_device) {
  if (on_device) { c->xcur = x; return VGPA_OK; }
  { int rc = ensure(c, &c->d_x, (size_t)c->B * c->len_x); if (rc) return rc; }
  HIP_TRY(c, hipMemcpyAsync(c->d_x, x, (size_t)c->B * c->len_x * sizeof(double), hipMemcpyHostToDevice, c->stream));
  c->xcur = c->d_x;
  return VGPA_OK;
}

// operator-level inputs arrive as separate [B][Np][D][D] / [B][Np][D] host arrays: pack them into the x layout
static int ingest_ab(vgpa_ctx* c, const double* lin_a, const double* off_b) {
  const size_t na = (size_t)c->Np * c->DD, nb = (size_t)c->Np * c->D;
  { int rc = ensure(c, &c->d_x, (size_t)c->B * c->len_x); if (rc) return rc; }
  c->xcur = c->d_x;
  if (lin_a) HIP_TRY(c, hipMemcpy2DAsync(c->d_x, c->len_x * sizeof(double), lin_a, na * sizeof(double), na * sizeof(double), c->B, hipMemcpyHostToDevice, c->stream));
  if (off_b) HIP_TRY(c, hipMemcpy2DAsync(c->d_x + na, c->len_x * sizeof(double), off_b, nb * sizeof(double), nb * sizeof(double), c->B, hipMemcpyHostToDevice, c->stream));
  return VGPA_OK;
}

static void prof_mark(vgpa_ctx* c, int i) {
  if (c->prof) (void)hipEventRecord(c->ev[i], c->stream);
}

static void prof_collect(vgpa_ctx* c) {
  if (!c->prof || !c->prof_pending) return;
  (void)hipEventSynchronize(c->ev[4]);
  for (int i = 0; i < 4; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) c->prof_ms[i] += ms;
  }
  c->prof_n += c->B;
  c->prof_pending = false;
}

// D <= 4: one lane per problem (any inputs) -- always at D = 1, from 512 problems at D = 2..4 (below that a problem per
// workgroup has the shorter latency: Lorenz-63 RK4 Np = 1001 forward 1.2 ms vs 1.6 ms; at 65536 problems 108 ms vs
// 6 ms).  VGPA_FLAG_FORCE_GENERIC keeps the workgroup-per-problem kernels.
static bool use_lane(vgpa_ctx* c) {
  // (the lane kernels address a wave's 64 problems with 32-bit byte offsets from a wave-uniform base)
  return c->D <= kMaxLaneD && !(c->cfg.flags & VGPA_FLAG_FORCE_GENERIC) && (c->D == 1 || c->B >= 512) && c->len_x < ((size_t)1 << 22);
}
// the fused lane pass (ode_small.hip::k_sweep_lane): forward kernel -> observations -> ONE kernel for the E_sde terms, the backward
// recursion, the gradient and F
static bool lane_fused(vgpa_ctx* c) {
  // (msT carries S_t as its lower triangle: a non-symmetric s0 / Sigma keeps the four-kernel path, which handles both halves literally)
  return use_lane(c) && c->full && sweep_lane_supported(c->cfg.model, c->D) && !(c->cfg.flags & VGPA_FLAG_MATERIALIZE) &&
         (c->D == 1 || c->sym_inputs);
}
// D = 2..4 below that: 16 lanes per problem, operands exchanged by ds_bpermute (ode_wave.hip)
static bool use_wave(vgpa_ctx* c) {
  return c->D >= 2 && c->D <= kMaxLaneD && !(c->cfg.flags & VGPA_FLAG_FORCE_GENERIC) && !use_lane(c);
}

// D <= 44 has two families of matrix-core stepping kernels: the symmetric-unit ones (two problems per CU, 4 waves each) win
// once there are more problems than CUs, the role-specialised ones (one problem per CU, 8 waves) below that and for one
// problem.  (D = 41 .. 44: one symmetric-unit workgroup per CU only -- its LDS -- so the role-specialised kernels stay.)
// 33 <= D <= 40 (round 3): the fragment-cover kernels win at every batch size -- a lone workgroup steps 4 % faster than the
// role-specialised pair (4.45 / 4.93 against 4.65 / 4.97 ms per forward / backward sweep of one problem), and the Q'' stream and
// the pipelined gradient assembly come with them; VGPA_ODE_KERNEL=pe in the environment keeps the role-specialised family there
// (comparison runs, tests).
static bool use_sym_units(vgpa_ctx* c) { return c->sym_units; }   // decided once in vgpa_create (pick_kernel_family)

static void pick_kernel_family(vgpa_ctx* c) {
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->cfg.device) != hipSuccess || n_cu <= 0) n_cu = 256;
  const char* fam = getenv("VGPA_ODE_KERNEL");
  const bool keep_pe = fam && !strcmp(fam, "pe");
  const int nb = (c->D + 3) / 4;
  // (D > 44 has no other matrix-core stepper: launch_ode_mfma always takes the symmetric-unit kernels there, so the context
  //  says so too and the energy kernel writes dEsde_dS as the upper triangle those kernels read)
  c->sym_units = (c->cfg.flags & VGPA_FLAG_SYM_UNITS) != 0 || (c->B > n_cu && nb <= 10) || ((nb == 9 || nb == 10) && !keep_pe) ||
                 (nb >= 12 && c->D <= kMaxSmallD);
}

static bool use_mfma(vgpa_ctx* c, bool fwd, bool sym) {
  return sym && !(c->cfg.flags & VGPA_FLAG_FORCE_GENERIC) && ode_mfma_supported(c->cfg.method, fwd, c->D);
}

static int ensure_ld_ws(vgpa_ctx* c) {
  if (c->d_ld_ws) return VGPA_OK;
  return dev_alloc(c, &c->d_ld_ws, (size_t)c->B * ld::ld_workspace_doubles(c->D));
}

// D > 64 with several problems per context: the per-stage kernels take the batch in grid.z; the drivers look every pointer's
// per-problem stride up by address (ld::BatchMap).  RAII: the map is thread-local state of large_d.hip.
struct LdLiteral {
  explicit LdLiteral(bool on_) : on(on_) { if (on) ld::ld_set_literal_products(true); }
  ~LdLiteral() { if (on) ld::ld_set_literal_products(false); }
  bool on;
};
struct LdBatch {
  explicit LdBatch(vgpa_ctx* c) : on(c->B > 1) {
    if (!on) return;
    ld::BatchMap m;
    m.nb = c->B;
    const size_t NpD = (size_t)c->Np * c->D, NpDD = (size_t)c->Np * c->DD;
    m.add(c->xcur, c->len_x);
    m.add(c->d_m, NpD); m.add(c->d_S, NpDD); m.add(c->d_lam, NpD); m.add(c->d_psi, NpDD);
    m.add(c->d_dEm, NpD); m.add(c->d_dEs, NpDD);
    m.add(c->d_jm, (size_t)c->M * c->D); m.add(c->d_jm_dense, NpD); m.add(c->d_js_dense, NpDD);
    m.add(c->d_ld_ws, ld::ld_workspace_doubles(c->D));
    ld::ld_set_batch(&m);
  }
  ~LdBatch() { if (on) ld::ld_set_batch(nullptr); }
  bool on;
};

static int run_fwd(vgpa_ctx* c, const double* m0, const double* S0, const double* Sigma, bool sym) {
  ld::use_library_gemm = (c->cfg.flags & VGPA_FLAG_LIBRARY_GEMM) != 0;
  c->ms_valid = true;                  // (every path below writes the [B][Np] arrays m / S)
  if (c->D > kMaxSmallD) {
    int rc = ensure_ld_ws(c);
    if (rc) return rc;
    LdBatch batch(c);
    LdLiteral literal(!sym);             // non-symmetric s0 / sigma: both products of the slope literally (ode_solver.py:60)
    hipError_t e = ld::ld_solve_fwd(c->cfg.method, c->cfg.dt, c->D, c->Np, ctx_A(c), ctx_b(c), m0, S0, Sigma, c->d_m, c->d_S,
                                    c->d_ld_ws, c->stream);
    if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "large-D forward sweep failed: %s", hipGetErrorString(e));
    return VGPA_OK;
  }
  OdeArgs a{};
  a.sym_units = use_sym_units(c) ? 1 : 0;
  a.D = c->D; a.Np = c->Np; a.batch = c->B; a.dt = c->cfg.dt;
  a.strideA = a.strideB = c->len_x;
  a.A = ctx_A(c); a.b = ctx_b(c); a.m0 = m0; a.S0 = S0; a.Sigma = Sigma; a.m = c->d_m; a.S = c->d_S;
  a.s_packed = c->s_packed ? 1 : 0;
  hipError_t e = use_lane(c) ? launch_ode_small(c->cfg.method, true, a, c->stream)
                 : use_wave(c) ? launch_ode_wave(c->cfg.method, true, a, c->stream)
                 : use_mfma(c, true, sym) ? launch_ode_mfma(c->cfg.method, true, a, c->stream)
                                          : launch_ode_generic(c->cfg.method, true, a, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "forward sweep launch failed: %s", hipGetErrorString(e));
  return VGPA_OK;
}

static bool grad_fused_ok(vgpa_ctx* c);
// g_fused: the backward kernel assembles the gradient into it (grad_fused_ok contexts; Psi_t is then not stored)
static int run_bwd(vgpa_ctx* c, bool dense_jumps, bool sym, double* g_fused = nullptr) {
  ld::use_library_gemm = (c->cfg.flags & VGPA_FLAG_LIBRARY_GEMM) != 0;
  int rc;
  if (g_fused && (dense_jumps || !grad_fused_ok(c) || !c->s_packed)) return fail(c, VGPA_ERR_STATE, "fused gradient assembly asked of a context without it");
  if ((rc = ensure(c, &c->d_psi, (size_t)c->B * c->Np * c->DD))) return rc;
  if ((rc = ensure(c, &c->d_dEs, (size_t)c->B * c->Np * c->DD))) return rc;
  c->psi_is_q = false;
  if (c->D > kMaxSmallD) {
    if ((rc = ensure_ld_ws(c))) return rc;
    LdBatch batch(c);
    LdLiteral literal(!sym);             // non-symmetric dEsde_ds / dEobs_ds (ode_solver.py:94)
    // operator-level calls bring dense jump arrays; the fused sweep uses the sparse ones (obs index on the host)
    hipError_t e = dense_jumps
        ? ld::ld_solve_bwd(c->cfg.method, c->cfg.dt, c->D, c->Np, ctx_A(c), c->d_dEm, c->d_dEs, c->d_jm_dense, c->d_js_dense,
                           c->d_lam, c->d_psi, c->d_ld_ws, c->stream)
        : ld::ld_solve_bwd(c->cfg.method, c->cfg.dt, c->D, c->Np, ctx_A(c), c->d_dEm, c->d_dEs, c->d_jm, c->d_jsc,
                           c->d_lam, c->d_psi, c->d_ld_ws, c->stream, c->h_obs_idx.data());
    if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "large-D backward sweep failed: %s", hipGetErrorString(e));
    return VGPA_OK;
  }
  OdeArgs a{};
  a.sym_units = use_sym_units(c) ? 1 : 0;
  a.D = c->D; a.Np = c->Np; a.batch = c->B; a.dt = c->cfg.dt;
  a.strideA = a.strideB = c->len_x;
  a.A = ctx_A(c); a.dEm = c->d_dEm; a.dEs = c->d_dEs; a.lam = c->d_lam; a.psi = c->d_psi;
  if (dense_jumps) { a.jm_dense = c->d_jm_dense; a.js_dense = c->d_js_dense; }
  else { a.obs_idx = c->d_obs_idx; a.jm_sparse = c->d_jm; a.js_const = c->des_packed ? c->d_jscp : c->d_jsc; a.n_obs = c->M; }
  a.ds_packed = c->des_packed ? 1 : 0;
  // fused sweeps on the fragment-cover kernels: Q''_t instead of Psi_t (the gradient assembly then does not read A_t; see
  // VGPA_FLAG_KEEP_PSI).  Same condition as the kernel choice below and as run_grad's matrix-core assembly.
  c->psi_is_q = !dense_jumps && a.sym_units && !use_lane(c) && !use_wave(c) && use_mfma(c, false, sym) && c->sigma_diag && c->isg_iso &&
                c->cfg.model == VGPA_MODEL_L96 && !(c->cfg.flags & VGPA_FLAG_KEEP_PSI) && sym_stores_q(c->cfg.method, c->D);
  a.q_on = c->psi_is_q ? 1 : 0;
  a.q_scale = c->isg0;
  if (g_fused) {                           // the gradient assembly on the kernel's helper waves (k_ode_sym, GF)
    if (!c->psi_is_q) return fail(c, VGPA_ERR_STATE, "fused gradient assembly: the backward kernel is not the Q'' one");
    a.grad_on = 1; a.g = g_fused; a.s_packed = 1;
    a.S = c->d_S; a.m = c->d_m; a.b = ctx_b(c); a.Ef = c->d_Ef; a.Am = c->d_Am;
    c->psi_is_q = false;                   // (nothing is stored in d_psi)
  }
  c->bwd_stored = !g_fused;
  hipError_t e = use_lane(c) ? launch_ode_small(c->cfg.method, false, a, c->stream)
                 : use_wave(c) ? launch_ode_wave(c->cfg.method, false, a, c->stream)
                 : use_mfma(c, false, sym) ? launch_ode_mfma(c->cfg.method, false, a, c->stream)
                                           : launch_ode_generic(c->cfg.method, false, a, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "backward sweep launch failed: %s", hipGetErrorString(e));
  return VGPA_OK;
}

static EnergyArgs energy_args(vgpa_ctx* c, double* edf, bool ds_upper = false) {
  EnergyArgs a{};
  a.ds_upper = ds_upper ? 1 : 0;
  a.s_packed = c->s_packed ? 1 : 0;
  a.model = c->cfg.model; a.D = c->D; a.Np = c->Np; a.batch = c->B; a.dt = c->cfg.dt;
  for (int i = 0; i < kMaxTheta; i++) a.theta[i] = c->theta[i];
  a.sigma1 = c->sigma1; a.isg = c->d_isg;
  a.strideA = a.strideB = c->len_x;
  a.A = ctx_A(c); a.b = ctx_b(c); a.m = c->d_m; a.S = c->d_S;
  a.e_t = c->d_et; a.Ef = c->d_Ef; a.Edf = edf; a.dEm = c->d_dEm; a.dEs = c->d_dEs; a.status = c->d_status;
  a.Am = (c->cfg.model == VGPA_MODEL_L96) ? c->d_Am : nullptr;
  a.hyp = c->hyp_on ? c->d_hyp : nullptr;
  return a;
}

static int ensure_lde_ws(vgpa_ctx* c) {
  if (!c->stream2 && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) {
    c->stream2 = nullptr;               // (lde_energy then runs on the one stream)
    (void)hipGetLastError();
  }
  if (c->d_lde_ws) return VGPA_OK;
  c->lde_nb = ld::lde_batch(c->D, c->lde_budget);
  if (c->lde_nb > c->Np) c->lde_nb = c->Np;
  return dev_alloc(c, &c->d_lde_ws, ld::lde_workspace_doubles(c->D, c->lde_nb));
}

static int run_energy(vgpa_ctx* c, double* edf, bool ds_upper = false, bool ds_packed = false) {
  c->des_upper = false;
  c->des_packed = false;
  { int rc = ensure(c, &c->d_dEs, (size_t)c->B * c->Np * c->DD); if (rc) return rc; }
  if (c->D > kMaxSmallD) {
    if (c->cfg.model != VGPA_MODEL_L96) return fail(c, VGPA_ERR_UNSUPPORTED, "large-D energy terms exist for Lorenz-96 only");
    int rc = ensure_lde_ws(c);
    if (rc) return rc;
    // the energy terms are batched over grid points already: problem by problem
    const size_t NpD = (size_t)c->Np * c->D, NpDD = (size_t)c->Np * c->DD;
    for (int p = 0; p < c->B; p++) {
      hipError_t e = ld::lde_energy(c->D, c->Np, c->theta[0], c->d_isg, ctx_A(c) + p * c->len_x, ctx_b(c) + p * c->len_x, c->d_m + p * NpD,
                                    c->d_S + p * NpDD, c->d_et + (size_t)p * c->Np, c->d_Ef + p * NpD, edf ? edf + p * NpDD : nullptr,
                                    c->d_dEm + p * NpD, c->d_dEs + p * NpDD, c->d_status + p, c->d_lde_ws, c->lde_nb, c->stream,
                                    c->hyp_on ? c->d_hyp + (size_t)p * c->Np * 2 * c->D : nullptr, c->stream2);
      if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "large-D energy failed: %s", hipGetErrorString(e));
    }
    return VGPA_OK;
  }
  c->des_upper = ds_upper && c->cfg.model == VGPA_MODEL_L96 && c->D >= 5;       // (the L96 kernels of 5 <= D <= 64 honour it)
  c->des_packed = c->des_upper && ds_packed && c->s_packed;                     // (k_energy_l96_r, the kernel that reads packed S_t, does)
  if (c->des_packed) c->des_upper = false;
  EnergyArgs a = energy_args(c, edf, c->des_upper);
  a.ds_packed = c->des_packed ? 1 : 0;
  hipError_t e = launch_energy(a, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "energy launch failed: %s", hipGetErrorString(e));
  return VGPA_OK;
}

static ObsArgs obs_args(vgpa_ctx* c) {
  ObsArgs a{};
  a.D = c->D; a.Np = c->Np; a.batch = c->B; a.n_obs = c->M; a.single = c->single ? 1 : 0;
  a.obs_t = c->d_obs_t; a.obs_y = c->d_obs_y; a.Q = c->d_Q; a.K = c->d_K; a.rinv_diag = c->d_rinv;
  a.obs_const = c->obs_const; a.m = c->d_m; a.S = c->d_S; a.jm_sparse = c->d_jm; a.eobs = c->d_eobs;
  a.diag = c->obs_diag ? 1 : 0; a.part = c->d_obs_part;
  a.s_packed = c->s_packed ? 1 : 0;
  return a;
}

static int run_reduce(vgpa_ctx* c) {
  ReduceArgs r{};
  r.Np = c->Np; r.batch = c->B; r.dt = c->cfg.dt; r.e0 = c->cfg.e0;
  r.pre = c->single ? 0.5 : 1.0; r.div = c->single ? c->sigma1 : 1.0;
  r.e_t = c->d_et; r.eobs = c->d_eobs; r.esde = c->d_esde; r.f = c->d_f;
  hipError_t e = launch_reduce(r, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "reduce launch failed: %s", hipGetErrorString(e));
  return VGPA_OK;
}

static int unpack_S(vgpa_ctx* c, const double** full);

static int run_grad(vgpa_ctx* c, double* g_dev) {
  if (c->D > kMaxSmallD) {
    int rc = ensure_lde_ws(c);
    if (rc) return rc;
    const size_t NpD = (size_t)c->Np * c->D, NpDD = (size_t)c->Np * c->DD;
    for (int p = 0; p < c->B; p++) {
      double* gp = g_dev + p * c->len_x;
      hipError_t e = ld::lde_grad(c->D, c->Np, c->cfg.dt, c->d_isg, ctx_A(c) + p * c->len_x, ctx_b(c) + p * c->len_x, c->d_m + p * NpD,
                                  c->d_S + p * NpDD, c->d_lam + p * NpD, c->d_psi + p * NpDD, c->d_Ef + p * NpD, gp, gp + NpDD, c->d_lde_ws,
                                  c->lde_nb, c->stream, c->sigma_diag ? nullptr : c->d_isig);
      if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "large-D gradient failed: %s", hipGetErrorString(e));
    }
    return VGPA_OK;
  }
  GradArgs a{};
  a.model = c->cfg.model; a.D = c->D; a.Np = c->Np; a.batch = c->B; a.sigma_diag = c->sigma_diag ? 1 : 0;
  a.dt = c->cfg.dt;
  for (int i = 0; i < kMaxTheta; i++) a.theta[i] = c->theta[i];
  a.strideA = a.strideB = c->len_x;
  a.isig = c->d_isig; a.A = ctx_A(c); a.b = ctx_b(c); a.m = c->d_m; a.S = c->d_S; a.lam = c->d_lam; a.psi = c->d_psi;
  a.Ef = c->d_Ef; a.Edf = nullptr; a.g = g_dev;
  a.psi_is_q = c->psi_is_q ? 1 : 0;
  a.s_packed = (c->s_packed && c->psi_is_q) ? 1 : 0;
  if (c->s_packed && !c->psi_is_q) {      // (an assembly kernel that wants S_t whole: behind VGPA_FETCH_PSIT, which recovers Psi_t in place and clears psi_is_q)
    const double* full = nullptr;
    int rc = unpack_S(c, &full);
    if (rc) return rc;
    a.S = full;
  }
  a.Am = c->d_Am;                                  // written by the L96 energy kernel of the same sweep (else null)
  a.scalar_product = (c->cfg.flags & VGPA_FLAG_FORCE_GENERIC) ? 1 : 0;
  hipError_t e = launch_grad(a, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "gradient launch failed: %s", hipGetErrorString(e));
  return VGPA_OK;
}

// ---- time-chunked large-D sweep (BASELINE config 4: Np x D x D arrays do not all fit in HBM) -------------------------
// Keeps x, S_t and the gradient (caller's buffer); Psi_t and dEsde_dS_t exist only for a chunk of ld_chunk + 1 grid
// points.  From the last grid point backwards, per chunk [t0, t1]:  energy terms of the chunk -> backward steps
// t1 .. t0+1 (Psi in the chunk buffer, lam_t in full) -> gradient of the grid points whose Psi_t is now final.
// g_dev == nullptr: energy integrand only (what F needs).  Same kernels, same per-grid-point arithmetic as the
// resident path, so the results are identical.
static int stream_pass(vgpa_ctx* c, double* g_dev) {
  ld::use_library_gemm = (c->cfg.flags & VGPA_FLAG_LIBRARY_GEMM) != 0;
  const int D = c->D, Np = c->Np, C = c->ld_chunk;
  const size_t DD = c->DD;
  int rc;
  if ((rc = ensure_lde_ws(c))) return rc;
  if ((rc = ensure_ld_ws(c))) return rc;
  if ((rc = ensure(c, &c->d_dEs_c, (size_t)(C + 1) * DD))) return rc;
  if (g_dev && (rc = ensure(c, &c->d_psi_c, (size_t)(C + 1) * DD))) return rc;
  const double *A = ctx_A(c), *b = ctx_b(c);
  hipStream_t st = c->stream;
  int t1 = Np - 1;
  if (g_dev) HIP_TRY(c, hipMemsetAsync(c->d_lam + (size_t)t1 * D, 0, sizeof(double) * D, st));
  bool first = true;
  while (true) {
    const int t0 = (t1 - C > 0) ? (t1 - C) : 0;
    const int n = t1 - t0 + 1;
    hipError_t e = ld::lde_energy(D, n, c->theta[0], c->d_isg, A + (size_t)t0 * DD, b + (size_t)t0 * D, c->d_m + (size_t)t0 * D,
                                  c->d_S + (size_t)t0 * DD, c->d_et + t0, c->d_Ef + (size_t)t0 * D, nullptr,
                                  c->d_dEm + (size_t)t0 * D, c->d_dEs_c, c->d_status, c->d_lde_ws, c->lde_nb, st, nullptr, c->stream2);
    if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "large-D energy failed: %s", hipGetErrorString(e));
    if (g_dev) {
      // Psi_{t1} sits in slot n-1: zero at the very end of the grid, else carried over from slot 0 of the previous chunk
      if (first) HIP_TRY(c, hipMemsetAsync(c->d_psi_c + (size_t)(n - 1) * DD, 0, sizeof(double) * DD, st));
      for (int t = t1; t > t0; t--) {
        const int k = t - t0;
        const int nobs = c->h_obs_idx[t - 1];
        e = ld::ld_bwd_step(c->cfg.method, c->cfg.dt, D, A + (size_t)t * DD, A + (size_t)(t - 1) * DD,
                            c->d_dEs_c + (size_t)k * DD, c->d_dEs_c + (size_t)(k - 1) * DD, c->d_dEm + (size_t)t * D,
                            c->d_dEm + (size_t)(t - 1) * D, c->d_psi_c + (size_t)k * DD, c->d_lam + (size_t)t * D,
                            c->d_psi_c + (size_t)(k - 1) * DD, c->d_lam + (size_t)(t - 1) * D,
                            nobs >= 0 ? c->d_jsc : nullptr, nobs >= 0 ? c->d_jm + (size_t)nobs * D : nullptr, c->d_ld_ws, st);
        if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "large-D backward step failed: %s", hipGetErrorString(e));
      }
      // gradient of (t0, t1] -- and of t0 itself once the grid start is reached
      const int g0 = (t0 == 0) ? 0 : t0 + 1;
      const int gn = t1 - g0 + 1;
      double* gA = g_dev + (size_t)g0 * DD;
      double* gB = g_dev + (size_t)Np * DD + (size_t)g0 * D;
      e = ld::lde_grad(D, gn, c->cfg.dt, c->d_isg, A + (size_t)g0 * DD, b + (size_t)g0 * D, c->d_m + (size_t)g0 * D,
                       c->d_S + (size_t)g0 * DD, c->d_lam + (size_t)g0 * D, c->d_psi_c + (size_t)(g0 - t0) * DD,
                       c->d_Ef + (size_t)g0 * D, gA, gB, c->d_lde_ws, c->lde_nb, st, c->sigma_diag ? nullptr : c->d_isig);
      if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "large-D gradient failed: %s", hipGetErrorString(e));
    }
    if (t0 == 0) break;
    if (g_dev) {   // Psi_{t0} becomes Psi_{t1} of the next chunk: slot 0 -> slot (t0 - t0_next)
      const int t0n = (t0 - C > 0) ? (t0 - C) : 0;
      HIP_TRY(c, hipMemcpyAsync(c->d_psi_c + (size_t)(t0 - t0n) * DD, c->d_psi_c, sizeof(double) * DD, hipMemcpyDeviceToDevice, st));
    }
    t1 = t0;
    first = false;
  }
  return VGPA_OK;
}

// fwd -> E_obs -> chunked [E_sde terms -> bwd -> gradient] -> F, for the streamed large-D context
static int enqueue_stream_sweep(vgpa_ctx* c, double* g_dev) {
  if (!c->full) return fail(c, VGPA_ERR_STATE, "context was created without m0/s0/observations (ODE-only)");
  int rc;
  prof_collect(c);
  HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t) * c->B, c->stream));
  c->s_packed = false;
  prof_mark(c, 0);
  if ((rc = run_fwd(c, c->d_m0, c->d_S0, c->d_Sigma, c->sym_inputs))) return rc;
  prof_mark(c, 1);
  hipError_t e = launch_obs(obs_args(c), c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "obs launch failed: %s", hipGetErrorString(e));
  prof_mark(c, 2);
  if ((rc = stream_pass(c, g_dev))) return rc;
  prof_mark(c, 3);
  if ((rc = run_reduce(c))) return rc;
  c->have_state = true;
  if (g_dev && c->prof) { prof_mark(c, 4); c->prof_pending = true; }
  return VGPA_OK;
}

// DIAGNOSTIC (tools/power_per_kernel.sh): VGPA_DIAG_REPEAT="<fwd|energy|bwd|grad>:<n>" launches that phase of the fused sweep n times
// instead of once -- every phase is a pure function of its inputs, so the results do not change -- to hold one kernel on the chip long
// enough for clock / power samples.  Read once; absent = 1.
static int diag_repeat(const char* phase) {
  static const std::string spec = [] { const char* e = getenv("VGPA_DIAG_REPEAT"); return std::string(e ? e : ""); }();
  const size_t colon = spec.find(':');
  if (colon == std::string::npos || spec.compare(0, colon, phase) != 0) return 1;
  const int n = atoi(spec.c_str() + colon + 1);
  return n > 1 ? n : 1;
}

// the fused lane pass over the cached (m, S): F (want_grad = false) or F and the gradient
static int run_lane_pass(vgpa_ctx* c, double* g_dev) {
  LaneSweepArgs q{};
  OdeArgs& a = q.o;
  a.D = c->D; a.Np = c->Np; a.batch = c->B; a.dt = c->cfg.dt;
  a.strideA = a.strideB = c->len_x;
  a.A = ctx_A(c); a.b = ctx_b(c); a.m = c->d_m; a.S = c->d_S;
  a.msT = c->d_msT; a.bpad = c->bpad; a.jmT = c->d_jmT;
  a.obs_idx = c->d_obs_idx; a.jm_sparse = c->d_jm; a.js_const = c->d_jsc; a.n_obs = c->M;
  q.model = c->cfg.model; q.want_grad = g_dev ? 1 : 0;
  for (int i = 0; i < kMaxTheta; i++) q.theta[i] = c->theta[i];
  q.sigma1 = c->sigma1;
  for (int i = 0; i < c->D; i++) q.isg[i] = c->h_isig[(size_t)i * c->D + i];
  for (size_t e = 0; e < c->DD; e++) q.isig[e] = c->h_isig[e];
  q.e0 = c->cfg.e0; q.pre = c->single ? 0.5 : 1.0; q.div = c->single ? c->sigma1 : 1.0;
  q.eobs = c->d_eobs; q.esde = c->d_esde; q.f = c->d_f; q.g = g_dev;
  hipError_t e = launch_sweep_lane(c->cfg.method, q, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "fused lane pass launch failed: %s", hipGetErrorString(e));
  return VGPA_OK;
}

// forward moments -> observation terms -> fused lane pass (F, and the gradient when g_dev is given)
static int enqueue_lane_sweep(vgpa_ctx* c, double* g_dev) {
  int rc;
  prof_collect(c);
  HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t) * c->B, c->stream));
  prof_mark(c, 0);
  if (!c->d_msT) {
    c->bpad = 64 * ((c->B + 63) / 64);
    if ((rc = dev_alloc(c, &c->d_msT, (size_t)c->Np * (c->D * (c->D + 1) / 2 + c->D) * c->bpad))) return rc;
    if ((rc = dev_alloc(c, &c->d_jmT, (size_t)(c->M > 0 ? c->M : 1) * c->D * c->bpad))) return rc;
  }
  {
    OdeArgs a{};
    a.D = c->D; a.Np = c->Np; a.batch = c->B; a.dt = c->cfg.dt;
    a.strideA = a.strideB = c->len_x;
    a.A = ctx_A(c); a.b = ctx_b(c); a.m0 = c->d_m0; a.S0 = c->d_S0; a.Sigma = c->d_Sigma; a.m = c->d_m; a.S = c->d_S;
    a.msT = c->d_msT; a.bpad = c->bpad;
    hipError_t ef = launch_ode_small(c->cfg.method, true, a, c->stream);
    if (ef != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "forward lane kernel launch failed: %s", hipGetErrorString(ef));
  }
  c->ms_valid = false;
  prof_mark(c, 1);
  hipError_t e = launch_obs_lane(obs_args(c), c->d_msT, c->bpad, c->d_jmT, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "obs launch failed: %s", hipGetErrorString(e));
  prof_mark(c, 2);
  if ((rc = run_lane_pass(c, g_dev))) return rc;
  prof_mark(c, 3);
  c->have_state = true;
  c->derived_valid = false;
  if (g_dev && c->prof) { prof_mark(c, 4); c->prof_pending = true; }
  return VGPA_OK;
}

// what vgpa_fetch wants of the arrays the fused lane pass never wrote: the separate kernels over the cached (m, S)
static int materialize_moments(vgpa_ctx* c) {
  if (c->ms_valid) return VGPA_OK;
  hipError_t e = launch_ms_untranspose(c->D, c->Np, c->B, c->bpad, c->d_msT, c->d_m, c->d_S, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "moment untranspose launch failed: %s", hipGetErrorString(e));
  c->ms_valid = true;
  return VGPA_OK;
}

static int materialize_derived(vgpa_ctx* c) {
  if (c->derived_valid) return VGPA_OK;
  int rc;
  if ((rc = materialize_moments(c))) return rc;
  {      // the separate backward kernel reads the jumps in the [B][M][D] layout: the observation kernel over the [B][Np] moments
    hipError_t e = launch_obs(obs_args(c), c->stream);
    if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "obs launch failed: %s", hipGetErrorString(e));
  }
  if ((rc = run_energy(c, nullptr, false))) return rc;
  if ((rc = run_bwd(c, false, c->sym_inputs))) return rc;
  c->derived_valid = true;
  return VGPA_OK;
}

// S_t as its packed lower triangle between the kernels of a fused sweep: exactly where the gradient assembly will be k_grad_mfma_q
// (the condition of run_bwd's Q'' stream: fragment-cover kernels, 33 <= D <= 40, RK2 / RK4, Sigma = sigma^2 I, Lorenz-96) and the
// energy terms come from k_energy_l96_r; VGPA_S_PACKED=0 in the environment keeps whole matrices (comparison runs)
static bool s_packed_ok(vgpa_ctx* c) {
  static const bool off = [] { const char* e = getenv("VGPA_S_PACKED"); return e && e[0] == '0'; }();
  return !off && c->sym_units && !use_lane(c) && !use_wave(c) && use_mfma(c, false, c->sym_inputs) && use_mfma(c, true, c->sym_inputs) &&
         c->sigma_diag && c->isg_iso && c->cfg.model == VGPA_MODEL_L96 && !(c->cfg.flags & (VGPA_FLAG_KEEP_PSI | VGPA_FLAG_FORCE_GENERIC)) &&
         sym_stores_q(c->cfg.method, c->D) && !c->hyp_on && c->D <= kMaxSmallD;
}

// the backward kernel assembles the gradient itself (OdeArgs::grad_on): wherever S_t is packed and the stepper's kernel can.  F-only
// evaluations of such a context skip the backward recursion altogether (F does not depend on it); gradient(x, eval_fun=False) runs it.
static bool grad_fused_ok(vgpa_ctx* c) {
  return s_packed_ok(c) && sym_fuses_grad(c->cfg.method, c->D) && c->d_Am != nullptr;
}
// ... from kFusedGradMinBatch problems on (VGPA_FUSED_GRAD=1 in the environment: always).  The third wave set costs the recursion
// ~0.4-0.5 ms per launch round (its matrix-core and vector-ALU instructions share the SIMDs' issue port with the product waves), the
// separate assembly ~7 us per problem: below ~70 problems the backward kernel followed by k_grad_mfma_q is the shorter way.
constexpr int kFusedGradMinBatch = 64;
static bool grad_fused_now(vgpa_ctx* c) {
  static const bool always = [] { const char* e = getenv("VGPA_FUSED_GRAD"); return e && e[0] == '1'; }();
  return grad_fused_ok(c) && c->s_packed && (always || c->B >= kFusedGradMinBatch);
}

// dEsde_dS between the energy kernel and the backward cover kernel as packed lower triangles: wherever S_t is packed (the same two
// kernels sit on either side); VGPA_DS_PACKED=0 in the environment keeps the upper triangles in whole matrices (comparison runs)
static bool ds_packed_ok(vgpa_ctx* c) {
  static const bool off = [] { const char* e = getenv("VGPA_DS_PACKED"); return e && e[0] == '0'; }();
  return !off && c->s_packed && c->d_jscp != nullptr;
}

// consumers that want S_t whole (vgpa_fetch, the operator-level kernels): the unpacked copy
static int unpack_S(vgpa_ctx* c, const double** full) {
  *full = c->d_S;
  if (!c->s_packed) return VGPA_OK;
  const size_t BN = (size_t)c->B * c->Np;
  int rc;
  if ((rc = ensure(c, &c->d_Sfull, BN * c->DD))) return rc;
  hipError_t e = launch_unpack_lower(BN, c->D, c->d_S, c->d_Sfull, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "unpack launch failed: %s", hipGetErrorString(e));
  *full = c->d_Sfull;
  return VGPA_OK;
}

// fwd -> E_obs -> E_sde terms -> bwd -> F     (VarGP.free_energy, variational.py:141-200)
static int enqueue_free_energy(vgpa_ctx* c) {
  if (c->stream_ld) return enqueue_stream_sweep(c, nullptr);
  if (!c->full) return fail(c, VGPA_ERR_STATE, "context was created without m0/s0/observations (ODE-only)");
  if (lane_fused(c)) return enqueue_lane_sweep(c, nullptr);
  int rc;
  prof_collect(c);
  c->s_packed = s_packed_ok(c);
  HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t) * c->B, c->stream));
  prof_mark(c, 0);
  for (int r = diag_repeat("fwd"); r > 0; r--)
    if ((rc = run_fwd(c, c->d_m0, c->d_S0, c->d_Sigma, c->sym_inputs))) return rc;
  prof_mark(c, 1);
  hipError_t e = launch_obs(obs_args(c), c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "obs launch failed: %s", hipGetErrorString(e));
  // (a symmetric-unit backward kernel reads the upper triangle of dEsde_dS only: the energy kernel writes nothing else then)
  const bool sym_bwd = use_sym_units(c) && !use_lane(c) && !use_wave(c) && use_mfma(c, false, c->sym_inputs) && c->D <= kMaxSmallD &&
                       !(c->cfg.flags & VGPA_FLAG_KEEP_PSI);
  for (int r = diag_repeat("energy"); r > 0; r--)
    if ((rc = run_energy(c, nullptr, sym_bwd, ds_packed_ok(c)))) return rc;
  prof_mark(c, 2);
  if (grad_fused_ok(c)) {                  // F needs no backward recursion; the gradient's comes with its assembly (finish_gradient)
    if ((rc = run_reduce(c))) return rc;
    c->bwd_stored = false;
    c->psi_is_q = false;
    c->have_state = true;
    c->derived_valid = true;
    return VGPA_OK;
  }
  for (int r = diag_repeat("bwd"); r > 0; r--)
    if ((rc = run_bwd(c, false, c->sym_inputs))) return rc;
  prof_mark(c, 3);
  if ((rc = run_reduce(c))) return rc;
  c->have_state = true;
  c->derived_valid = true;
  return VGPA_OK;
}

static int check_status(vgpa_ctx* c) {
  std::vector<int32_t> st(c->B);
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->d_status, sizeof(int32_t) * c->B, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int p = 0; p < c->B; p++)
    if (st[p] & 1)
      return fail(c, VGPA_ERR_NOT_PD, "problem %d: marginal covariance S_t is not positive definite "
                  "(reference: LinAlgError from chol_inv, variational.py:380)", p);
  return VGPA_OK;
}

// =====================================================================================================
extern "C" {

#ifdef VGPA_EXPERIMENTS
int vgpa_abi_version(void) { return VGPA_ABI_VERSION | VGPA_ABI_DIAGNOSTIC_BUILD; }
#else
int vgpa_abi_version(void) { return VGPA_ABI_VERSION; }
#endif

int vgpa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* vgpa_last_error(const vgpa_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void vgpa_destroy(vgpa_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (void* p : c->allocs) (void)hipFree(p);
  for (void* p : c->user_allocs) (void)hipFree(p);
  if (c->h_coef) (void)hipHostFree(c->h_coef);
  if (c->h_fs) (void)hipHostFree(c->h_fs);
  for (auto& e : c->ev_coef) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int vgpa_create(vgpa_ctx** out, const vgpa_config* cfg) {
  if (!out || !cfg) return fail(nullptr, VGPA_ERR_ARG, "null argument");
  *out = nullptr;
  if (cfg->abi_version != VGPA_ABI_VERSION) return fail(nullptr, VGPA_ERR_ARG, "ABI version mismatch: %d != %d", cfg->abi_version, VGPA_ABI_VERSION);
  if (!(cfg->dt > 0.0)) return fail(nullptr, VGPA_ERR_ARG, "Discrete time step should be strictly positive -> %g.", cfg->dt);
  if (cfg->method < VGPA_ODE_EULER || cfg->method > VGPA_ODE_RK4) return fail(nullptr, VGPA_ERR_ARG, "Integration method is unknown -> %d.", cfg->method);
  if (cfg->model < VGPA_MODEL_NONE || cfg->model > VGPA_MODEL_L96) return fail(nullptr, VGPA_ERR_ARG, "Unknown stochastic model -> %d", cfg->model);
  if (cfg->dim_d < 1 || cfg->n_pts < 2 || cfg->batch < 1 || cfg->n_obs < 0) return fail(nullptr, VGPA_ERR_ARG, "bad sizes: D=%d Np=%d batch=%d M=%d", cfg->dim_d, cfg->n_pts, cfg->batch, cfg->n_obs);
  const bool single = (cfg->model == VGPA_MODEL_OU || cfg->model == VGPA_MODEL_DW) ||
                      (cfg->model == VGPA_MODEL_NONE && cfg->dim_d == 1);
  if (single && cfg->dim_d != 1) return fail(nullptr, VGPA_ERR_ARG, "1-D model with D=%d", cfg->dim_d);
  if (cfg->model == VGPA_MODEL_L63 && cfg->dim_d != 3) return fail(nullptr, VGPA_ERR_ARG, "Lorenz-63 needs D=3, got %d", cfg->dim_d);
  if (cfg->model == VGPA_MODEL_L96 && cfg->dim_d < 4) return fail(nullptr, VGPA_ERR_ARG, "Insufficient state vector dimensions: %d", cfg->dim_d);
  if (cfg->dim_d > kMaxSmallD && cfg->model != VGPA_MODEL_NONE && cfg->model != VGPA_MODEL_L96)
    return fail(nullptr, VGPA_ERR_UNSUPPORTED, "D=%d > %d is built for the ODE operators (model NONE) and for Lorenz-96 only",
                cfg->dim_d, kMaxSmallD);
  const int need_theta = (cfg->model == VGPA_MODEL_L63) ? 3 : (cfg->model == VGPA_MODEL_NONE ? 0 : 1);
  if (cfg->n_theta != need_theta || (need_theta > 0 && !cfg->theta)) return fail(nullptr, VGPA_ERR_ARG, "model needs %d drift parameter(s)", need_theta);
  if (!cfg->sigma) return fail(nullptr, VGPA_ERR_ARG, "sigma is required");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, VGPA_ERR_DEVICE, "no HIP device is visible (libvgpa_hip has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, VGPA_ERR_DEVICE, "device %d out of range (%d visible)", cfg->device, ndev);

  vgpa_ctx* c = new vgpa_ctx();
  c->cfg = *cfg;
  c->D = cfg->dim_d; c->Np = cfg->n_pts; c->B = cfg->batch; c->M = cfg->n_obs;
  c->DD = (size_t)c->D * c->D;
  c->len_x = (size_t)c->Np * c->DD + (size_t)c->Np * c->D;
  c->single = single;
  for (int i = 0; i < cfg->n_theta; i++) c->theta[i] = cfg->theta[i];
  c->full = cfg->model != VGPA_MODEL_NONE && cfg->m0 && cfg->s0 && (cfg->n_obs == 0 || (cfg->obs_t && cfg->obs_y && cfg->obs_noise));
  const int D = c->D;
  const size_t DD = c->DD;
  int rc = VGPA_OK;
#define TRY(expr) do { rc = (expr); if (rc != VGPA_OK) { g_create_error = c->err; vgpa_destroy(c); return rc; } } while (0)
#define HTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(nullptr, VGPA_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); vgpa_destroy(c); return VGPA_ERR_DEVICE; } } while (0)
  HTRY(hipSetDevice(cfg->device));
  HTRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  pick_kernel_family(c);
  // (phase events: no system-scope fence behind them -- nothing on the host reads device memory at a phase boundary; with the default
  //  flags five events cost a batched Ornstein-Uhlenbeck step 0.4 of its 1.4 ms)
  for (auto& e : c->ev) HTRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));

  // ---- host-side constants -------------------------------------------------------------------
  std::vector<double> sigma(cfg->sigma, cfg->sigma + DD), isig(DD, 0.0), isg(D, 0.0);
  if (single) {
    if (!(sigma[0] > 0.0)) { fail(nullptr, VGPA_ERR_ARG, "The diffusion noise value: %g, should be strictly positive.", sigma[0]); vgpa_destroy(c); return VGPA_ERR_ARG; }
    c->sigma1 = sigma[0]; isig[0] = 1.0 / sigma[0]; isg[0] = isig[0];
  } else if (cfg->model == VGPA_MODEL_NONE) {
    for (int i = 0; i < D; i++) { isig[(size_t)i * D + i] = 1.0; isg[i] = 1.0; }   // ODE-only: Sigma^-1 is not used
  } else {
    c->sigma_diag = true;
    for (int i = 0; i < D && c->sigma_diag; i++)
      for (int j = 0; j < D; j++)
        if (i != j && sigma[(size_t)i * D + j] != 0.0) { c->sigma_diag = false; break; }
    if (c->sigma_diag) {   // chol_inv of a diagonal matrix: C = diag(1/sqrt(s)), C^T C = diag(C_ii * C_ii)
      for (int i = 0; i < D; i++) {
        const double sii = sigma[(size_t)i * D + i];
        if (!(sii > 0.0)) { fail(nullptr, VGPA_ERR_NOT_PD, "Noise matrix is not positive definite."); vgpa_destroy(c); return VGPA_ERR_NOT_PD; }
        const double ci = 1.0 / std::sqrt(sii);
        isig[(size_t)i * D + i] = ci * ci;
        isg[i] = ci * ci;
      }
    } else {
      if (!host_spd_inverse(D, sigma.data(), isig.data(), nullptr)) { fail(nullptr, VGPA_ERR_NOT_PD, "Noise matrix is not positive definite."); vgpa_destroy(c); return VGPA_ERR_NOT_PD; }
      for (int i = 0; i < D; i++) isg[i] = isig[(size_t)i * D + i];
    }
  }
  c->sym_inputs = is_symmetric(sigma.data(), D) && (!cfg->s0 || is_symmetric(cfg->s0, D));

  const size_t BN = (size_t)c->B * c->Np;
  TRY(dev_alloc(c, &c->d_m, BN * D));
  TRY(dev_alloc(c, &c->d_S, BN * DD));
  TRY(dev_alloc(c, &c->d_Ef, BN * D));
  if (cfg->model == VGPA_MODEL_L96 && D <= kMaxSmallD) TRY(dev_alloc(c, &c->d_Am, BN * D));
  TRY(dev_alloc(c, &c->d_dEm, BN * D));
  TRY(dev_alloc(c, &c->d_lam, BN * D));
  TRY(dev_alloc(c, &c->d_et, BN));
  TRY(dev_alloc(c, &c->d_eobs, (size_t)c->B));
  TRY(dev_alloc(c, &c->d_esde, (size_t)c->B));
  TRY(dev_alloc(c, &c->d_f, (size_t)c->B));
  TRY(dev_alloc(c, &c->d_status, (size_t)c->B));
  TRY(dev_alloc(c, &c->d_jm, (size_t)c->B * (c->M > 0 ? c->M : 1) * D));
  TRY(dev_alloc(c, &c->d_Sigma, DD));
  TRY(dev_alloc(c, &c->d_isig, DD));
  TRY(dev_alloc(c, &c->d_isg, (size_t)D));
  TRY(dev_alloc(c, &c->d_m0, (size_t)D));
  TRY(dev_alloc(c, &c->d_S0, DD));
  TRY(dev_alloc(c, &c->d_op_m0, (size_t)D));
  TRY(dev_alloc(c, &c->d_op_S0, DD));
  TRY(dev_alloc(c, &c->d_op_Sigma, DD));
  TRY(dev_alloc(c, &c->d_obs_idx, (size_t)c->Np));
  TRY(dev_alloc(c, &c->d_obs_t, (size_t)(c->M > 0 ? c->M : 1)));
  TRY(dev_alloc(c, &c->d_obs_y, (size_t)(c->M > 0 ? c->M : 1) * D));
  TRY(dev_alloc(c, &c->d_Q, DD));
  TRY(dev_alloc(c, &c->d_K, DD));
  TRY(dev_alloc(c, &c->d_rinv, (size_t)D));
  TRY(dev_alloc(c, &c->d_jsc, DD));
  HTRY(hipMemsetAsync(c->d_status, 0, sizeof(int32_t) * c->B, c->stream));
  HTRY(hipMemsetAsync(c->d_eobs, 0, sizeof(double) * c->B, c->stream));
  HTRY(hipMemsetAsync(c->d_jm, 0, sizeof(double) * (size_t)c->B * (c->M > 0 ? c->M : 1) * D, c->stream));
  HTRY(hipMemsetAsync(c->d_jsc, 0, sizeof(double) * DD, c->stream));

  TRY(upload(c, c->d_Sigma, sigma.data(), DD));
  c->h_isig = isig;
  TRY(upload(c, c->d_isig, isig.data(), DD));
  TRY(upload(c, c->d_isg, isg.data(), (size_t)D));
  c->isg0 = isg[0];
  c->isg_iso = true;
  for (int i = 1; i < D; i++) c->isg_iso = c->isg_iso && isg[i] == isg[0];
  if (cfg->m0) TRY(upload(c, c->d_m0, cfg->m0, (size_t)D));
  if (cfg->s0) TRY(upload(c, c->d_S0, cfg->s0, DD));

  std::vector<int32_t> obs_idx(c->Np, -1);
  std::vector<double> Q(DD, 0.0), K(DD, 0.0), rinv(D, 0.0), jsc(DD, 0.0);
  if (c->M > 0 && cfg->obs_t && cfg->obs_y && cfg->obs_noise) {
    for (int n = 0; n < c->M; n++) {
      const int64_t tn = cfg->obs_t[n];
      if (tn < 0 || tn >= c->Np || (n > 0 && tn <= cfg->obs_t[n - 1])) { fail(nullptr, VGPA_ERR_ARG, "obs_t must be strictly increasing indices in [0, Np)"); vgpa_destroy(c); return VGPA_ERR_ARG; }
      obs_idx[tn] = n;
    }
    if (single) {
      const double r = cfg->obs_noise[0];
      if (!(r > 0.0)) { fail(nullptr, VGPA_ERR_NOT_PD, "observation noise must be positive"); vgpa_destroy(c); return VGPA_ERR_NOT_PD; }
      const double h = cfg->obs_h ? cfg->obs_h[0] : 1.0;
      Q[0] = 1.0 / r; K[0] = h; rinv[0] = 1.0 / r; jsc[0] = 0.5 / r;
      c->obs_const = 0.5 * c->M * (std::log(2.0 * M_PI) + std::log(r));
    } else {
      std::vector<double> Rinv(DD, 0.0), H(DD, 0.0), T(DD);
      double logdet = 0.0;
      // fast path: diagonal R and H = I (no O(D^3) host work at large D).  An explicitly passed identity counts as
      // "no operator" (the reference's Likelihood materialises np.eye(d) when the operator is None, likelihood.py:33-40).
      bool h_identity = true;
      if (cfg->obs_h)
        for (int i = 0; i < D && h_identity; i++)
          for (int j = 0; j < D; j++)
            if (cfg->obs_h[(size_t)i * D + j] != (i == j ? 1.0 : 0.0)) { h_identity = false; break; }
      bool r_diag = h_identity;
      for (int i = 0; i < D && r_diag; i++)
        for (int j = 0; j < D; j++)
          if (i != j && cfg->obs_noise[(size_t)i * D + j] != 0.0) { r_diag = false; break; }
      if (r_diag) {
        for (int i = 0; i < D; i++) {
          const double rii = cfg->obs_noise[(size_t)i * D + i];
          if (!(rii > 0.0)) { fail(nullptr, VGPA_ERR_NOT_PD, "observation noise matrix is not positive definite"); vgpa_destroy(c); return VGPA_ERR_NOT_PD; }
          const double ci = 1.0 / std::sqrt(rii);
          const double ri = ci * ci;
          Q[(size_t)i * D + i] = ri; K[(size_t)i * D + i] = ri; jsc[(size_t)i * D + i] = 0.5 * ri; rinv[i] = ri;
          logdet += std::log(std::sqrt(rii));
        }
        logdet *= 2.0;
        c->obs_const = c->M * (D * std::log(2.0 * M_PI) + logdet);
      } else if (!host_spd_inverse(D, cfg->obs_noise, Rinv.data(), &logdet)) { fail(nullptr, VGPA_ERR_NOT_PD, "observation noise matrix is not positive definite"); vgpa_destroy(c); return VGPA_ERR_NOT_PD; }
      if (!r_diag) {
      if (cfg->obs_h && !h_identity) H.assign(cfg->obs_h, cfg->obs_h + DD); else for (int i = 0; i < D; i++) H[(size_t)i * D + i] = 1.0;
      host_matmul(D, H.data(), Rinv.data(), T.data(), false, false);      // H R^-1
      host_matmul(D, T.data(), H.data(), Q.data(), false, true);          // H R^-1 H^T
      host_matmul(D, H.data(), Rinv.data(), T.data(), true, false);       // H^T R^-1
      host_matmul(D, T.data(), H.data(), K.data(), false, true);          // H^T R^-1 H^T
      host_matmul(D, T.data(), H.data(), jsc.data(), false, false);       // H^T R^-1 H
      for (auto& v : jsc) v *= 0.5;
      for (int i = 0; i < D; i++) rinv[i] = Rinv[(size_t)i * D + i];
      c->obs_const = c->M * (D * std::log(2.0 * M_PI) + logdet);
      }
      c->sym_inputs = c->sym_inputs && is_symmetric(jsc.data(), D);
      c->obs_diag = r_diag;
      if (D > kMaxSmallD) TRY(dev_alloc(c, &c->d_obs_part, (size_t)c->B * c->M));   // one workgroup per observation
    }
    TRY(upload(c, c->d_obs_t, cfg->obs_t, (size_t)c->M));
    TRY(upload(c, c->d_obs_y, cfg->obs_y, (size_t)c->M * D));
  }
  c->h_obs_idx = obs_idx;
  TRY(upload(c, c->d_obs_idx, obs_idx.data(), (size_t)c->Np));
  if (D > kMaxSmallD && c->full && cfg->model == VGPA_MODEL_L96) {
    // resident large-D sweep: S, dEsde_dS, Psi next to the caller's x and gradient.  Stream when asked to, or when
    // that does not fit into what is free on the device now.
    size_t free_b = 0, total_b = 0;
    HTRY(hipMemGetInfo(&free_b, &total_b));
    const double need = 8.0 * (double)BN * (double)DD * 4.0;        // dEs + Psi + the caller's x and g still to come
    c->stream_ld = (cfg->flags & VGPA_FLAG_STREAM_LARGE_D) != 0 || need > 0.9 * (double)free_b;
    if (c->stream_ld && c->B > 1) {      // (everything allocated so far goes back: this fires exactly when memory is short)
      const int nb_ = c->B;
      vgpa_destroy(c);
      return fail(nullptr, VGPA_ERR_UNSUPPORTED, "the time-chunked large-D sweep holds one problem (a batch of %d does not fit resident)", nb_);
    }
    c->lde_budget = std::fmin(16.0e9, std::fmax(1.0e9, 0.05 * (double)free_b));   // workspace of the batched energy terms
    c->ld_chunk = ld::lde_batch(D, c->lde_budget) - 1;   // a chunk evaluates ld_chunk + 1 grid points: exactly one energy batch
    if (c->ld_chunk > c->Np - 1) c->ld_chunk = c->Np - 1;
    if (c->ld_chunk < 1) c->ld_chunk = 1;
  }
  TRY(upload(c, c->d_Q, Q.data(), DD));
  TRY(upload(c, c->d_K, K.data(), DD));
  TRY(upload(c, c->d_rinv, rinv.data(), (size_t)D));
  TRY(upload(c, c->d_jsc, jsc.data(), DD));
  {      // the same (symmetric) matrix as a packed lower triangle, for the backward kernels that read a packed dEsde_dS stream
    std::vector<double> jp((size_t)DD, 0.0);
    for (int r = 0; r < D; r++)
      for (int q = 0; q <= r; q++) jp[(size_t)r * (r + 1) / 2 + q] = jsc[(size_t)r * D + q];
    TRY(dev_alloc(c, &c->d_jscp, DD));
    TRY(upload(c, c->d_jscp, jp.data(), DD));
  }
  HTRY(hipStreamSynchronize(c->stream));
#undef TRY
#undef HTRY
  *out = c;
  return VGPA_OK;
}

int vgpa_synchronize(vgpa_ctx* c) {
  if (!c) return VGPA_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VGPA_OK;
}

void* vgpa_stream(vgpa_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---- operator level ---------------------------------------------------------------------------------
int vgpa_solve_fwd(vgpa_ctx* c, const double* lin_a, const double* off_b, const double* m0, const double* s0,
                   const double* sigma, double* mt, double* st) {
  if (!c || !lin_a || !off_b || !m0 || !s0 || !sigma || !mt || !st) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t BN = (size_t)c->B * c->Np;
  int rc;
  if ((rc = ingest_ab(c, lin_a, off_b))) return rc;
  if ((rc = upload(c, c->d_op_m0, m0, (size_t)c->D))) return rc;
  if ((rc = upload(c, c->d_op_S0, s0, c->DD))) return rc;
  if ((rc = upload(c, c->d_op_Sigma, sigma, c->DD))) return rc;
  const bool sym = is_symmetric(s0, c->D) && is_symmetric(sigma, c->D);
  c->s_packed = false;                 // operator-level results are whole matrices
  if ((rc = run_fwd(c, c->d_op_m0, c->d_op_S0, c->d_op_Sigma, sym))) return rc;
  if ((rc = download(c, mt, c->d_m, BN * c->D))) return rc;
  if ((rc = download(c, st, c->d_S, BN * c->DD))) return rc;
  c->have_state = false;
  return vgpa_synchronize(c);
}

int vgpa_solve_bwd(vgpa_ctx* c, const double* lin_a, const double* desde_dm, const double* desde_ds,
                   const double* deobs_dm, const double* deobs_ds, double* lam, double* psi) {
  if (!c || !lin_a || !desde_dm || !desde_ds || !deobs_dm || !deobs_ds || !lam || !psi) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t BN = (size_t)c->B * c->Np;
  int rc;
  if (!c->d_jm_dense) {
    if ((rc = dev_alloc(c, &c->d_jm_dense, BN * c->D))) return rc;
    if ((rc = dev_alloc(c, &c->d_js_dense, BN * c->DD))) return rc;
  }
  if ((rc = ingest_ab(c, lin_a, nullptr))) return rc;
  if ((rc = upload(c, c->d_dEm, desde_dm, BN * c->D))) return rc;
  if ((rc = ensure(c, &c->d_dEs, BN * c->DD))) return rc;
  if ((rc = ensure(c, &c->d_psi, BN * c->DD))) return rc;
  if ((rc = upload(c, c->d_dEs, desde_ds, BN * c->DD))) return rc;
  c->des_upper = false;                  // the caller's array is complete
  c->des_packed = false;
  if ((rc = upload(c, c->d_jm_dense, deobs_dm, BN * c->D))) return rc;
  if ((rc = upload(c, c->d_js_dense, deobs_ds, BN * c->DD))) return rc;
  const bool sym = stack_symmetric(desde_ds, BN, c->D) && stack_symmetric(deobs_ds, BN, c->D);
  if ((rc = run_bwd(c, true, sym))) return rc;
  if ((rc = download(c, lam, c->d_lam, BN * c->D))) return rc;
  if ((rc = download(c, psi, c->d_psi, BN * c->DD))) return rc;
  c->have_state = false;
  return vgpa_synchronize(c);
}

// dEsde/dtheta and dEsde/dSigma of <model>.energy (ornstein_uhlenbeck.py:222-226, double_well.py:250-254,
// lorenz_63.py:329-342, lorenz_96.py:421-434): the kernels emit the per-grid-point integrands, one trapezoid per
// component reduces them, the O(D^3) scaling by Sigma^-1 is finished on the host.
int vgpa_energy_full(vgpa_ctx* c, const double* lin_a, const double* off_b, const double* mt, const double* st,
                     double* esde_out, double* efx, double* edf, double* desde_dm, double* desde_ds,
                     double* desde_dth, double* desde_dsig) {
  if (!c || !lin_a || !off_b || !mt || !st) return fail(c, VGPA_ERR_ARG, "null argument");
  if ((desde_dth != nullptr) != (desde_dsig != nullptr)) return fail(c, VGPA_ERR_ARG, "dEsde_dth and dEsde_dsig come together");
  const bool hyper = desde_dth != nullptr;
  if (hyper && c->cfg.model == VGPA_MODEL_NONE) return fail(c, VGPA_ERR_STATE, "context has no stochastic model");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const int D = c->D;
  const int H = c->single ? 1 : 2 * D;
  const size_t BN = (size_t)c->B * c->Np;
  int rc;
  if (edf && !c->d_Edf && (rc = dev_alloc(c, &c->d_Edf, BN * c->DD))) return rc;
  if (hyper && !c->d_hyp) {
    if ((rc = dev_alloc(c, &c->d_hyp, BN * H))) return rc;
    if ((rc = dev_alloc(c, &c->d_hypT, (size_t)c->B * H))) return rc;
  }
  if ((rc = ingest_ab(c, lin_a, off_b))) return rc;
  c->ms_valid = true;
  c->s_packed = false;
  if ((rc = upload(c, c->d_m, mt, BN * c->D))) return rc;
  if ((rc = upload(c, c->d_S, st, BN * c->DD))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->d_status, 0, sizeof(int32_t) * c->B, c->stream));
  c->hyp_on = hyper;
  rc = run_energy(c, edf ? c->d_Edf : nullptr);
  c->hyp_on = false;
  if (rc) return rc;
  if ((rc = run_reduce(c))) return rc;
  if (hyper) {
    hipError_t e = launch_trapz_multi(c->d_hyp, c->Np, H, c->B, c->cfg.dt, c->d_hypT, c->stream);
    if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "trapezoid launch failed: %s", hipGetErrorString(e));
  }
  if ((rc = check_status(c))) return rc;
  std::vector<double> T(hyper ? (size_t)c->B * H : 0), esde(c->B);
  if ((rc = download(c, esde.data(), c->d_esde, esde.size()))) return rc;
  if (hyper && (rc = download(c, T.data(), c->d_hypT, T.size()))) return rc;
  if (efx && (rc = download(c, efx, c->d_Ef, BN * c->D))) return rc;
  if (edf && (rc = download(c, edf, c->d_Edf, BN * c->DD))) return rc;
  if (desde_dm && (rc = download(c, desde_dm, c->d_dEm, BN * c->D))) return rc;
  if (desde_ds && (rc = download(c, desde_ds, c->d_dEs, BN * c->DD))) return rc;
  c->have_state = false;
  if ((rc = vgpa_synchronize(c))) return rc;
  if (esde_out) for (int p = 0; p < c->B; p++) esde_out[p] = esde[p];
  if (!hyper) return VGPA_OK;
  for (int p = 0; p < c->B; p++) {
    const double* Tp = T.data() + (size_t)p * H;
    if (c->single) {
      desde_dth[p] = (c->cfg.model == VGPA_MODEL_DW ? 4.0 : 1.0) * Tp[0] / c->sigma1;
      desde_dsig[p] = -esde[p] / c->sigma1;
      continue;
    }
    const double* is = c->h_isig.data();
    for (int i = 0; i < D; i++) desde_dth[(size_t)p * D + i] = is[(size_t)i * D + i] * Tp[i];
    double* out = desde_dsig + (size_t)p * D * D;          // -0.5 * Sigma^-1 diag(v) Sigma^-1
    for (int i = 0; i < D; i++)
      for (int j = 0; j < D; j++) {
        double sacc = 0.0;
        for (int k = 0; k < D; k++) sacc += (is[(size_t)i * D + k] * Tp[D + k]) * is[(size_t)k * D + j];
        out[(size_t)i * D + j] = -0.5 * sacc;
      }
  }
  return VGPA_OK;
}

int vgpa_energy(vgpa_ctx* c, const double* lin_a, const double* off_b, const double* mt, const double* st,
                double* esde, double* efx, double* edf, double* desde_dm, double* desde_ds) {
  return vgpa_energy_full(c, lin_a, off_b, mt, st, esde, efx, edf, desde_dm, desde_ds, nullptr, nullptr);
}

int vgpa_obs_energy(vgpa_ctx* c, const double* mt, const double* st, double* eobs, double* deobs_dm, double* deobs_ds) {
  if (!c || !mt || !st) return fail(c, VGPA_ERR_ARG, "null argument");
  if (!c->cfg.obs_t && c->M > 0) return fail(c, VGPA_ERR_STATE, "context has no observations");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t BN = (size_t)c->B * c->Np;
  int rc;
  c->ms_valid = true;
  c->s_packed = false;
  if ((rc = upload(c, c->d_m, mt, BN * c->D))) return rc;
  if ((rc = upload(c, c->d_S, st, BN * c->DD))) return rc;
  ObsArgs a = obs_args(c);
  hipError_t e = launch_obs(a, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "obs launch failed: %s", hipGetErrorString(e));
  if (eobs && (rc = download(c, eobs, c->d_eobs, (size_t)c->B))) return rc;
  if (deobs_dm || deobs_ds) {
    if (!c->d_jm_dense) {
      if ((rc = dev_alloc(c, &c->d_jm_dense, BN * c->D))) return rc;
      if ((rc = dev_alloc(c, &c->d_js_dense, BN * c->DD))) return rc;
    }
    HIP_TRY(c, hipMemsetAsync(c->d_jm_dense, 0, sizeof(double) * BN * c->D, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_js_dense, 0, sizeof(double) * BN * c->DD, c->stream));
    e = launch_obs_dense(a, c->d_jsc, c->d_jm_dense, c->d_js_dense, c->stream);
    if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "obs dense launch failed: %s", hipGetErrorString(e));
    if (deobs_dm && (rc = download(c, deobs_dm, c->d_jm_dense, BN * c->D))) return rc;
    if (deobs_ds && (rc = download(c, deobs_ds, c->d_js_dense, BN * c->DD))) return rc;
  }
  c->have_state = false;
  return vgpa_synchronize(c);
}

// ---- fused objective --------------------------------------------------------------------------------
int vgpa_free_energy(vgpa_ctx* c, const double* x, double* f) {
  if (!c || !x || !f) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = ingest_x(c, x, false))) return rc;
  if ((rc = enqueue_free_energy(c))) return rc;
  return vgpa_fetch_f(c, f);
}

int vgpa_free_energy_dev(vgpa_ctx* c, const double* x_dev, double* f_host) {
  if (!c || !x_dev || !f_host) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = ingest_x(c, x_dev, true))) return rc;
  if ((rc = enqueue_free_energy(c))) return rc;
  return vgpa_fetch_f(c, f_host);
}

int vgpa_fetch_f(vgpa_ctx* c, double* f_host) {
  if (!c || !f_host) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  // F and the status words through one pinned block, one synchronisation (a pageable destination makes the copy synchronous by itself,
  // and check_status is a second round trip: ~0.3 ms per call of a batched context, 15 % of an Ornstein-Uhlenbeck step)
  const size_t B = (size_t)c->B;
  if (!c->h_fs && hipHostMalloc(&c->h_fs, B * (sizeof(double) + sizeof(int32_t))) != hipSuccess) {
    c->h_fs = nullptr;
    (void)hipGetLastError();
    int rc;
    if ((rc = download(c, f_host, c->d_f, B))) return rc;
    return check_status(c);
  }
  double* hf = static_cast<double*>(c->h_fs);
  int32_t* hs = reinterpret_cast<int32_t*>(hf + B);
  HIP_TRY(c, hipMemcpyAsync(hf, c->d_f, B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(hs, c->d_status, B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::memcpy(f_host, hf, B * sizeof(double));
  for (size_t p = 0; p < B; p++)
    if (hs[p] & 1)
      return fail(c, VGPA_ERR_NOT_PD, "problem %d: marginal covariance S_t is not positive definite "
                  "(reference: LinAlgError from chol_inv, variational.py:380)", (int)p);
  return VGPA_OK;
}

static int finish_gradient(vgpa_ctx* c, double* g_dev);

// F and the gradient in one go (df(x, eval_fun=True)): the streamed context folds both into one chunked pass
static int enqueue_sweep(vgpa_ctx* c, double* g_dev) {
  if (c->stream_ld) return enqueue_stream_sweep(c, g_dev);
  if (c->full && lane_fused(c)) return enqueue_lane_sweep(c, g_dev);
  int rc = enqueue_free_energy(c);
  return rc ? rc : finish_gradient(c, g_dev);
}

static int finish_gradient(vgpa_ctx* c, double* g_dev) {
  if (c->stream_ld) {            // cached state = (m, S): the chunked pass recomputes the energy terms on its way back
    int rc = stream_pass(c, g_dev);
    if (rc == VGPA_OK && c->prof) { prof_mark(c, 4); c->prof_pending = true; }
    return rc;
  }
  if (lane_fused(c) && !c->derived_valid) {      // gradient(x, eval_fun=False) behind a fused F: the pass again, now with the recursion
    int rc = run_lane_pass(c, g_dev);
    if (rc == VGPA_OK && c->prof) { prof_mark(c, 4); c->prof_pending = true; }
    return rc;
  }
  int rc = VGPA_OK;
  if (grad_fused_now(c)) {                 // backward recursion + gradient assembly in one kernel (phase "bwd"; "grad" is empty)
    for (int r = diag_repeat("bwd"); r > 0 && rc == VGPA_OK; r--) rc = run_bwd(c, false, c->sym_inputs, g_dev);
    prof_mark(c, 3);
    if (rc == VGPA_OK && c->prof) { prof_mark(c, 4); c->prof_pending = true; }
    return rc;
  }
  if (!c->bwd_stored) {                    // (F-only evaluation before: the recursion now, with Q''_t for the assembly kernel)
    for (int r = diag_repeat("bwd"); r > 0 && rc == VGPA_OK; r--) rc = run_bwd(c, false, c->sym_inputs);
    prof_mark(c, 3);
  }
  for (int r = diag_repeat("grad"); r > 0 && rc == VGPA_OK; r--) rc = run_grad(c, g_dev);
  if (rc == VGPA_OK && c->prof) { prof_mark(c, 4); c->prof_pending = true; }
  return rc;
}

int vgpa_gradient(vgpa_ctx* c, const double* x_or_null, double* g) {
  if (!c || !g) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = ensure(c, &c->d_g, (size_t)c->B * c->len_x))) return rc;
  if (x_or_null) {
    if ((rc = ingest_x(c, x_or_null, false))) return rc;
    if ((rc = enqueue_sweep(c, c->d_g))) return rc;
  } else if (!c->have_state) {
    return fail(c, VGPA_ERR_STATE, "gradient(x, eval_fun=False) needs the state cached by a previous free_energy");
  } else if ((rc = finish_gradient(c, c->d_g))) return rc;
  if ((rc = download(c, g, c->d_g, (size_t)c->B * c->len_x))) return rc;
  return check_status(c);
}

int vgpa_sweep(vgpa_ctx* c, const double* x, double* f, double* g) {
  if (!c || !x || !f || !g) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = ensure(c, &c->d_g, (size_t)c->B * c->len_x))) return rc;
  if ((rc = ingest_x(c, x, false))) return rc;
  if ((rc = enqueue_sweep(c, c->d_g))) return rc;
  if ((rc = download(c, g, c->d_g, (size_t)c->B * c->len_x))) return rc;
  return vgpa_fetch_f(c, f);
}

int vgpa_sweep_enqueue(vgpa_ctx* c, const double* x_dev, double* g_dev) {
  if (!c || !x_dev || !g_dev) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = ingest_x(c, x_dev, true))) return rc;
  return enqueue_sweep(c, g_dev);
}

int vgpa_sweep_dev(vgpa_ctx* c, const double* x_dev, double* f_host, double* g_dev) {
  if (!c || !f_host) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = vgpa_sweep_enqueue(c, x_dev, g_dev))) return rc;
  return vgpa_fetch_f(c, f_host);
}

int vgpa_energy_parts(vgpa_ctx* c, double* e0, double* esde, double* eobs) {
  if (!c) return VGPA_ERR_ARG;
  if (!c->have_state) return fail(c, VGPA_ERR_STATE, "no cached state");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if (e0) for (int p = 0; p < c->B; p++) e0[p] = c->cfg.e0;
  if (esde && (rc = download(c, esde, c->d_esde, (size_t)c->B))) return rc;
  if (eobs && (rc = download(c, eobs, c->d_eobs, (size_t)c->B))) return rc;
  return vgpa_synchronize(c);
}

int vgpa_fetch(vgpa_ctx* c, int which, double* out) {
  if (!c || !out) return fail(c, VGPA_ERR_ARG, "null argument");
  if (!c->have_state) return fail(c, VGPA_ERR_STATE, "no cached state: call free_energy first");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t BN = (size_t)c->B * c->Np;
  int rc = VGPA_OK;
  if ((rc = materialize_moments(c))) return rc;
  if (which != VGPA_FETCH_MT && which != VGPA_FETCH_ST && which != VGPA_FETCH_EDF && (rc = materialize_derived(c))) return rc;
  if ((which == VGPA_FETCH_LAMT || which == VGPA_FETCH_PSIT) && !c->bwd_stored && !c->stream_ld && (rc = run_bwd(c, false, c->sym_inputs))) return rc;
  switch (which) {
    case VGPA_FETCH_MT: rc = download(c, out, c->d_m, BN * c->D); break;
    case VGPA_FETCH_ST: {
      const double* full = nullptr;
      if ((rc = unpack_S(c, &full))) return rc;
      rc = download(c, out, full, BN * c->DD);
      break;
    }
    case VGPA_FETCH_LAMT: rc = download(c, out, c->d_lam, BN * c->D); break;
    case VGPA_FETCH_PSIT:
      if (!c->d_psi || c->stream_ld) return fail(c, VGPA_ERR_UNSUPPORTED, "Psi_t is not kept by the time-chunked large-D sweep");
      if (c->psi_is_q) {               // recover Psi_t = (Sigma^-1 A_t - Q''_t) / 2 in place: from here on d_psi holds Psi_t again
        hipError_t e = launch_psi_from_q(c->B, c->Np, c->D, c->len_x, ctx_A(c), c->d_isg, c->d_psi, c->stream);
        if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "Psi_t recovery launch failed: %s", hipGetErrorString(e));
        c->psi_is_q = false;
      }
      rc = download(c, out, c->d_psi, BN * c->DD); break;
    case VGPA_FETCH_EFX: rc = download(c, out, c->d_Ef, BN * c->D); break;
    case VGPA_FETCH_DESDE_DM: rc = download(c, out, c->d_dEm, BN * c->D); break;
    case VGPA_FETCH_DESDE_DS:
      if (!c->d_dEs || c->stream_ld) return fail(c, VGPA_ERR_UNSUPPORTED, "dEsde_dS is not kept by the time-chunked large-D sweep");
      if (c->des_packed) {             // whole matrices into the scratch copy the unpacked S_t uses too (the packed stream stays as it is)
        if ((rc = ensure(c, &c->d_Sfull, BN * c->DD))) return rc;
        hipError_t e = launch_unpack_lower(BN, c->D, c->d_dEs, c->d_Sfull, c->stream);
        if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "unpack launch failed: %s", hipGetErrorString(e));
        rc = download(c, out, c->d_Sfull, BN * c->DD);
        break;
      }
      if (c->des_upper) {
        hipError_t e = launch_mirror_upper(BN, c->D, c->d_dEs, c->stream);
        if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "mirror launch failed: %s", hipGetErrorString(e));
        c->des_upper = false;
      }
      rc = download(c, out, c->d_dEs, BN * c->DD); break;
    case VGPA_FETCH_ESDE_T: rc = download(c, out, c->d_et, BN); break;
    case VGPA_FETCH_EDF: {
      if (!c->d_Edf && (rc = dev_alloc(c, &c->d_Edf, BN * c->DD))) return rc;
      EnergyArgs a = energy_args(c, c->d_Edf);
      hipError_t e = launch_edf(a, c->stream);
      if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "edf launch failed: %s", hipGetErrorString(e));
      rc = download(c, out, c->d_Edf, BN * c->DD);
      break;
    }
    default: return fail(c, VGPA_ERR_ARG, "unknown fetch selector %d", which);
  }
  if (rc) return rc;
  return vgpa_synchronize(c);
}

int vgpa_gradient_dev(vgpa_ctx* c, double* g_dev) {
  if (!c || !g_dev) return fail(c, VGPA_ERR_ARG, "null argument");
  if (!c->have_state) return fail(c, VGPA_ERR_STATE, "gradient(x, eval_fun=False) needs the state cached by a previous free_energy");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = finish_gradient(c, g_dev))) return rc;
  return check_status(c);
}

// The *_dev entry points consume the caller's x in place: the cached state refers to that memory until the next
// evaluation.  A caller about to free or overwrite it says so here; gradient(x, eval_fun=False) then fails loudly
// instead of reading freed memory.
int vgpa_release_x(vgpa_ctx* c) {
  if (!c) return VGPA_ERR_ARG;
  if (c->xcur != c->d_x) { c->xcur = nullptr; c->have_state = false; }
  return VGPA_OK;
}

static int vec_scratch(vgpa_ctx* c, uint64_t seglen) {
  const size_t need = (size_t)c->B * (3 + (size_t)vec_blocks_per_seg((long long)seglen, c->B));
  if (c->vec_scratch_n >= need) return VGPA_OK;
  c->vec_scratch_n = (size_t)c->B * (3 + 256);
  return dev_alloc(c, &c->d_vec_scratch, c->vec_scratch_n);
}
static int vec_reduce_host(vgpa_ctx* c, int mode, const double* a, const double* b, uint64_t seglen, double* out) {
  if (!c || !a || !out || seglen == 0) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = vec_scratch(c, seglen))) return rc;
  double* red = c->d_vec_scratch + 2 * (size_t)c->B;    // [0,2B) holds the axpby coefficients
  hipError_t e = vec_reduce(mode, a, b, c->B, (long long)seglen, red, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "vector reduction failed: %s", hipGetErrorString(e));
  HIP_TRY(c, hipMemcpyAsync(out, red, sizeof(double) * c->B, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VGPA_OK;
}
int vgpa_vec_dot(vgpa_ctx* c, const double* a, const double* b, uint64_t seglen, double* out) {
  if (!b) return fail(c, VGPA_ERR_ARG, "null argument");
  return vec_reduce_host(c, 0, a, b, seglen, out);
}
int vgpa_vec_absmax(vgpa_ctx* c, const double* a, uint64_t seglen, double* out) { return vec_reduce_host(c, 1, a, nullptr, seglen, out); }
int vgpa_vec_asum(vgpa_ctx* c, const double* a, uint64_t seglen, double* out) { return vec_reduce_host(c, 2, a, nullptr, seglen, out); }
int vgpa_vec_axpby(vgpa_ctx* c, uint64_t seglen, const double* alpha, const double* x, const double* beta, const double* y, double* out) {
  if (!c || !alpha || !x || !out || seglen == 0 || ((y != nullptr) != (beta != nullptr))) return fail(c, VGPA_ERR_ARG, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = vec_scratch(c, seglen))) return rc;
  // The caller's coefficient arrays may be gone when this returns: copy them into a pinned ring slot first.  A slot is
  // reused only after the upload that read it has completed (its event); the device-side slots are stream-ordered
  // behind the previous axpby.
  const size_t B = (size_t)c->B;
  if (!c->h_coef) {
    HIP_TRY(c, hipHostMalloc((void**)&c->h_coef, sizeof(double) * 2 * B * vgpa_ctx::kCoefSlots, hipHostMallocDefault));
    for (auto& ev : c->ev_coef) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  const unsigned slot = c->coef_next % vgpa_ctx::kCoefSlots;
  if (c->coef_next >= (unsigned)vgpa_ctx::kCoefSlots) HIP_TRY(c, hipEventSynchronize(c->ev_coef[slot]));
  c->coef_next++;
  double* h = c->h_coef + (size_t)slot * 2 * B;
  std::memcpy(h, alpha, sizeof(double) * B);
  if (beta) std::memcpy(h + B, beta, sizeof(double) * B);
  HIP_TRY(c, hipMemcpyAsync(c->d_vec_scratch, h, sizeof(double) * (beta ? 2 * B : B), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev_coef[slot], c->stream));
  hipError_t e = vec_axpby(c->B, (long long)seglen, c->d_vec_scratch, x, c->d_vec_scratch + c->B, y, out, c->stream);
  if (e != hipSuccess) return fail(c, VGPA_ERR_DEVICE, "axpby failed: %s", hipGetErrorString(e));
  return VGPA_OK;
}

int vgpa_set_option(vgpa_ctx* c, int option, int64_t value) {
  if (!c) return VGPA_ERR_ARG;
  if (option == VGPA_OPT_LD_CHUNK) {
    if (value < 1) return fail(c, VGPA_ERR_ARG, "chunk must be >= 1");
    if (value > c->Np - 1) value = c->Np > 1 ? c->Np - 1 : 1;
    if (c->d_dEs_c || c->d_psi_c) return fail(c, VGPA_ERR_STATE, "the chunk buffers exist already: set the option before the first sweep");
    c->ld_chunk = (int)value;
    return VGPA_OK;
  }
  return fail(c, VGPA_ERR_ARG, "unknown option %d", option);
}

// E0 = KL(q0||p0) is constant in x but not in the prior: the reference recomputes it on every free_energy call
// (variational.py:185) from kl0.mu0 / kl0.tau0, which are plain attributes.  The host mirror passes the current value.
int vgpa_set_prior_energy(vgpa_ctx* c, double e0) {
  if (!c) return VGPA_ERR_ARG;
  c->cfg.e0 = e0;
  return VGPA_OK;
}

int vgpa_is_streaming(vgpa_ctx* c) { return (c && c->stream_ld) ? 1 : 0; }

// ---- raw device memory ------------------------------------------------------------------------------
int vgpa_dev_alloc(vgpa_ctx* c, uint64_t bytes, void** out) {
  if (!c || !out) return VGPA_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipMalloc(out, bytes ? bytes : 8));
  c->user_allocs.push_back(*out);          // whatever the caller does not return is freed with the context
  return VGPA_OK;
}
int vgpa_dev_free(vgpa_ctx* c, void* ptr) {
  if (!c) return VGPA_ERR_ARG;
  if (!ptr) return VGPA_OK;
  // (may run from a garbage collector at any point of the calling thread: its current device is put back)
  int prev = -1;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (ptr == (const void*)c->xcur) { c->xcur = nullptr; c->have_state = false; }
  for (size_t i = 0; i < c->user_allocs.size(); i++)
    if (c->user_allocs[i] == ptr) { c->user_allocs[i] = c->user_allocs.back(); c->user_allocs.pop_back(); break; }
  HIP_TRY(c, hipFree(ptr));
  return VGPA_OK;
}
int vgpa_memcpy_h2d(vgpa_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c) return VGPA_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VGPA_OK;
}
int vgpa_memcpy_d2h(vgpa_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c) return VGPA_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return VGPA_OK;
}

// ---- device memory without a context (the row-sharded driver's callers: vgpa_amd/large_d.py keeps its buffers here) ----------------
// Plain hipMalloc / hipFree / synchronous hipMemcpy on `device`; the calling thread's current device is put back.
namespace {
struct DeviceScope {
  int prev = -1; bool ok = false;
  explicit DeviceScope(int device) { (void)hipGetDevice(&prev); ok = hipSetDevice(device) == hipSuccess; }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace
int vgpa_device_alloc(int device, uint64_t bytes, void** out) {
  if (!out) return VGPA_ERR_ARG;
  DeviceScope sc(device);
  if (!sc.ok) return VGPA_ERR_DEVICE;
  return hipMalloc(out, bytes ? bytes : 8) == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
}
int vgpa_device_free(int device, void* ptr) {
  if (!ptr) return VGPA_OK;
  DeviceScope sc(device);
  if (!sc.ok) return VGPA_ERR_DEVICE;
  return hipFree(ptr) == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
}
int vgpa_device_memcpy(int device, void* dst, const void* src, uint64_t bytes, int kind) {
  if ((!dst || !src) && bytes) return VGPA_ERR_ARG;
  if (kind < 1 || kind > 3) return VGPA_ERR_ARG;
  DeviceScope sc(device);
  if (!sc.ok) return VGPA_ERR_DEVICE;
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
  if (bytes == 0) return VGPA_OK;
  // (every stream of the library is non-blocking: the null-stream copy below does not wait for them by itself)
  if (kind != 1 && hipDeviceSynchronize() != hipSuccess) return VGPA_ERR_DEVICE;
  if (hipMemcpy(dst, src, bytes, k) != hipSuccess) return VGPA_ERR_DEVICE;
  return hipDeviceSynchronize() == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
}

// ---- profiling ----------------------------------------------------------------------------------------
int vgpa_profile_begin(vgpa_ctx* c) {
  if (!c) return VGPA_ERR_ARG;
  c->prof = true; c->prof_pending = false; c->prof_n = 0;
  for (auto& v : c->prof_ms) v = 0.0;
  return VGPA_OK;
}
int vgpa_profile_end(vgpa_ctx* c, double* fwd_ms, double* energy_ms, double* bwd_ms, double* grad_ms, int64_t* n_sweeps) {
  if (!c) return VGPA_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  c->prof = false;
  if (fwd_ms) *fwd_ms = c->prof_ms[0];
  if (energy_ms) *energy_ms = c->prof_ms[1];
  if (bwd_ms) *bwd_ms = c->prof_ms[2];
  if (grad_ms) *grad_ms = c->prof_ms[3];
  if (n_sweeps) *n_sweeps = c->prof_n;
  return VGPA_OK;
}

}  // extern "C"
