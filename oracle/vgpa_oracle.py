"""
oracle/vgpa_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy CPU restatement of the reference's forward-backward variational
smoothing sweep (vrettasm/VGPA).  It exists only so that `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg can check / time
the HIP path against the reference's arithmetic on a machine where the reference
itself is not available.  Nothing under `vgpa_amd/` imports this module.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function below
against fixtures under `tests/golden/` that were produced by importing the real
reference in the build container (`tools/gen_golden.py`); agreement is <= 1e-12
relative.  The reference's own unit tests do not cover this path.

Every function cites the reference file:line it restates (paths relative to the
reference repository root).  Known reference quirks are
reproduced on purpose and flagged `Q<n>`.

Two modes for the Lorenz-96 energy:
  * faithful=True  : same per-step operations as the reference, including the
                     unscented-transform covariance that both callers discard
                     and the 2D+1 dense solves  ("the numpy CPU path").
  * faithful=False : dead work removed, same results to rounding ("lean").
"""
from dataclasses import dataclass, field

import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))


# --------------------------------------------------------------------------- #
#  Problem description (plain data; mirrors what Simulation.setup assembles,
#  src/var_bayes/simulation.py:92-178).
# --------------------------------------------------------------------------- #
@dataclass
class Problem:
    model: str                     # "OU" | "DW" | "L63" | "L96"
    method: str                    # "euler" | "heun" | "rk2" | "rk4"
    dt: float
    theta: np.ndarray              # scalar (OU, DW, L96) or (3,) (L63)
    sigma: np.ndarray              # scalar (1-D) or (D, D)
    m0: np.ndarray
    s0: np.ndarray
    mu0: np.ndarray
    tau0: np.ndarray
    obs_t: np.ndarray              # int64 (M,)
    obs_y: np.ndarray              # (M,) or (M, D)
    obs_noise: np.ndarray          # scalar or (D, D)
    n_pts: int
    dim_d: int
    obs_h: np.ndarray = None       # observation operator, identity when None
    extras: dict = field(default_factory=dict)

    @property
    def single_dim(self):
        return self.model in ("OU", "DW")

    @property
    def inverse_sigma(self):
        if self.single_dim:
            return 1.0 / self.sigma
        return chol_inv(self.sigma)[0]

    @classmethod
    def from_fixture(cls, z):
        model = str(z["model"])
        single = model in ("OU", "DW")
        sigma = float(z["sigma"]) if single else np.asarray(z["sigma"], dtype=float)
        return cls(model=model, method=str(z["method"]).lower(), dt=float(z["dt"]),
                   theta=(float(z["theta"]) if np.ndim(z["theta"]) == 0 else np.asarray(z["theta"], dtype=float)),
                   sigma=sigma,
                   m0=(float(z["m0"]) if single else np.asarray(z["m0"])),
                   s0=(float(z["s0"]) if single else np.asarray(z["s0"])),
                   mu0=(float(z["mu0"]) if single else np.asarray(z["mu0"])),
                   tau0=(float(z["tau0"]) if single else np.asarray(z["tau0"])),
                   obs_t=np.asarray(z["obs_t"], dtype=np.int64), obs_y=np.asarray(z["obs_y"]),
                   obs_noise=(float(z["obs_noise"]) if single else np.asarray(z["obs_noise"])),
                   n_pts=int(np.asarray(z["time_window"]).size),
                   dim_d=(1 if single else int(np.asarray(z["m0"]).size)))

    def split(self, x):
        """x -> (A, b); src/var_bayes/variational.py:153-162."""
        n, d = self.n_pts, self.dim_d
        if d == 1:
            return x[:n], x[n:]
        return x[:n * d * d].reshape(n, d, d), x[n * d * d:].reshape(n, d)


# --------------------------------------------------------------------------- #
#  Small linear-algebra helpers: src/numerics/utilities.py
# --------------------------------------------------------------------------- #
def chol_inv(x):
    """utilities.py:203-237.  Returns (x^-1, inverse of the lower Cholesky factor)."""
    x = np.asarray(x)
    if x.ndim == 0:
        return 1.0 / x, 1.0 / np.sqrt(x)
    c_inv = np.linalg.solve(np.linalg.cholesky(x), np.eye(x.shape[0]))
    return c_inv.T.dot(c_inv), c_inv


def log_det(x):
    """utilities.py:68-105 (Cholesky based; also used on a non-symmetric product, Q5)."""
    x = np.asarray(x)
    if x.ndim == 0:
        return np.log(x)
    if x.ndim == 1:
        x = np.diag(x)
    return 2.0 * np.sum(np.log(np.linalg.cholesky(x).diagonal()))


def _trapz0(fx, dx):
    """scipy.integrate.trapezoid(fx, dx=dx, axis=0) restated: sum(dx*(f[1:]+f[:-1])/2)."""
    fx = np.asarray(fx)
    return np.sum(dx * (fx[1:] + fx[:-1]) / 2.0, axis=0)


def my_trapz(fx, dx=1.0, obs_t=None):
    """utilities.py:144-201: trapezoid rule summed piecewise between observation indices."""
    if obs_t is None:
        return _trapz0(fx, dx)
    total, first = 0.0, 0
    for k, last in enumerate(obs_t):
        total += _trapz0(fx[first:last + 1], dx)
        first = obs_t[k]
    if first != fx.shape[0] - 1:
        total += _trapz0(fx[first:], dx)
    return total


def ut_approx(fun, x_bar, x_cov, *args, faithful=True):
    """
    Unscented transform, utilities.py:239-310.  kappa = 1.05*D (:271); sigma points are the
    ROWS [m; m + U; m - U] with U = chol_lower((D+kappa) S)^T (:275,283-288); on LinAlgError the
    factor is chol(S o I)^T, unscaled (:279).  The covariance (:302-306) is returned only in
    faithful mode (both callers discard it, Q8).
    """
    x_bar, x_cov = np.asarray(x_bar), np.asarray(x_cov)
    d = x_bar.size
    m_pts = 2 * d + 1
    kappa = 1.05 * d
    try:
        root = np.linalg.cholesky((d + kappa) * x_cov).T
    except np.linalg.LinAlgError:
        root = np.linalg.cholesky(x_cov * np.eye(d)).T
    chi = np.concatenate((x_bar[np.newaxis, :], x_bar + root, x_bar - root))
    w = np.full((1, m_pts), 1.0 / (2.0 * (d + kappa)))
    w[0, 0] = kappa / (d + kappa)
    y = fun(chi, *args)
    y_bar = w.dot(y).ravel()
    if not faithful:
        return y_bar, None
    w_m = np.eye(m_pts) - np.tile(w, (m_pts, 1))
    q_mat = w_m.dot(np.diag(w.ravel())).dot(w_m.T)
    return y_bar, y.T.dot(q_mat).dot(y)


# --------------------------------------------------------------------------- #
#  ODE right-hand sides: src/numerics/ode_solver.py:31-95
# --------------------------------------------------------------------------- #
def f_m(m, a, b, single):
    return -(a * m) + b if single else -a.dot(m) + b                      # ode_solver.py:44


def f_s(s, a, sn, single):
    return -(2.0 * a * s) + sn if single else -a.dot(s) - s.dot(a.T) + sn  # ode_solver.py:60


def f_lam(g, a, lam, single):
    return -g + (lam * a) if single else -g + lam.dot(a.T)                # ode_solver.py:77 (Q3)


def f_psi(g, a, psi, single):
    return -g + (2.0 * psi * a) if single else -g + psi.dot(a) + a.T.dot(psi)  # ode_solver.py:94


def _alloc(n, d, single):
    if single:
        return np.zeros(n), np.zeros(n)
    return np.zeros((n, d)), np.zeros((n, d, d))


def _mid_padded(z):
    """Mid-points 0.5*(z[k]+z[k+1]) zero-padded back to Np rows (runge_kutta4.py:148-173)."""
    mid = 0.5 * (z[0:-1] + z[1:])
    return np.concatenate((mid, np.zeros((1,) + z.shape[1:])), axis=0)


def solve_fwd(method, dt, single, lin_a, off_b, m0, s0, sigma):
    """
    Forward moment ODE (m_t, S_t).  euler.py:27-92, heun.py:28-111, runge_kutta2.py:25-102,
    runge_kutta4.py:25-113.
    """
    n = off_b.shape[0]
    d = 1 if single else off_b.shape[1]
    mt, st = _alloc(n, d, single)
    mt[0], st[0] = m0, s0
    h = 0.5 * dt
    if method in ("rk2", "rk4"):
        a_mid = 0.5 * (lin_a[0:-1] + lin_a[1:])
        b_mid = 0.5 * (off_b[0:-1] + off_b[1:])
    for k in range(n - 1):
        ak, bk, mk, sk = lin_a[k], off_b[k], mt[k], st[k]
        if method == "euler":
            mt[k + 1] = mk + f_m(mk, ak, bk, single) * dt
            st[k + 1] = sk + f_s(sk, ak, sigma, single) * dt
        elif method == "heun":
            ap, bp = lin_a[k + 1], off_b[k + 1]
            p = f_m(mk, ak, bk, single)
            c = f_m(mk + p * dt, ap, bp, single)
            mt[k + 1] = mk + h * (p + c)
            p = f_s(sk, ak, sigma, single)
            c = f_s(sk + p * dt, ap, sigma, single)
            st[k + 1] = sk + h * (p + c)
        elif method == "rk2":
            mt[k + 1] = mk + dt * f_m(mk + h * f_m(mk, ak, bk, single), a_mid[k], b_mid[k], single)
            # Q2: the predictor passes S_k in the place of A_k (runge_kutta2.py:96).
            st[k + 1] = sk + dt * f_s(sk + h * f_s(sk, sk, sigma, single), a_mid[k], sigma, single)
        elif method == "rk4":
            am, bm = a_mid[k], b_mid[k]
            k1 = f_m(mk, ak, bk, single)
            k2 = f_m(mk + h * k1, am, bm, single)
            k3 = f_m(mk + h * k2, am, bm, single)
            k4 = f_m(mk + dt * k3, lin_a[k + 1], off_b[k + 1], single)
            mt[k + 1] = mk + dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0
            l1 = f_s(sk, ak, sigma, single)
            l2 = f_s(sk + h * l1, am, sigma, single)
            l3 = f_s(sk + h * l2, am, sigma, single)
            l4 = f_s(sk + dt * l3, lin_a[k + 1], sigma, single)
            st[k + 1] = sk + dt * (l1 + 2.0 * (l2 + l3) + l4) / 6.0
        else:
            raise ValueError(f"unknown integration method: {method}")
    return mt, st


def solve_bwd(method, dt, single, lin_a, de_dm, de_ds, jump_m, jump_s):
    """
    Backward Lagrange ODE (lam_t, Psi_t) with observation jumps added after the step (Q9).
    euler.py:94-154, heun.py:113-190, runge_kutta2.py:104-194, runge_kutta4.py:115-211.
    """
    n = de_dm.shape[0]
    d = 1 if single else de_dm.shape[1]
    lam, psi = _alloc(n, d, single)
    h = 0.5 * dt
    if method in ("rk2", "rk4"):
        a_mid, em_mid, es_mid = _mid_padded(lin_a), _mid_padded(de_dm), _mid_padded(de_ds)
    for t in range(n - 1, 0, -1):
        at, lt, pt = lin_a[t], lam[t], psi[t]
        if method == "euler":
            lam[t - 1] = lt - f_lam(de_dm[t], at, lt, single) * dt + jump_m[t - 1]
            psi[t - 1] = pt - f_psi(de_ds[t], at, pt, single) * dt + jump_s[t - 1]
        elif method == "heun":
            ak = lin_a[t - 1]
            p = f_lam(de_dm[t], at, lt, single)
            c = f_lam(de_dm[t - 1], ak, lt - p * dt, single)
            lam[t - 1] = lt - h * (p + c) + jump_m[t - 1]
            p = f_psi(de_ds[t], at, pt, single)
            c = f_psi(de_ds[t - 1], ak, pt - p * dt, single)
            psi[t - 1] = pt - h * (p + c) + jump_s[t - 1]
        elif method == "rk2":
            ak, em, es = a_mid[t - 1], em_mid[t - 1], es_mid[t - 1]
            lk = lt - h * f_lam(de_dm[t], at, lt, single)
            lam[t - 1] = lt - dt * f_lam(em, ak, lk, single) + jump_m[t - 1]
            pk = pt - h * f_psi(de_ds[t], at, pt, single)
            psi[t - 1] = pt - dt * f_psi(es, ak, pk, single) + jump_s[t - 1]
        elif method == "rk4":
            ak, em, es = a_mid[t - 1], em_mid[t - 1], es_mid[t - 1]
            k1 = f_lam(de_dm[t], at, lt, single)
            k2 = f_lam(em, ak, lt - h * k1, single)
            k3 = f_lam(em, ak, lt - h * k2, single)
            k4 = f_lam(de_dm[t - 1], lin_a[t - 1], lt - dt * k3, single)
            lam[t - 1] = lt - dt * (k1 + 2.0 * (k2 + k3) + k4) / 6.0 + jump_m[t - 1]
            l1 = f_psi(de_ds[t], at, pt, single)
            l2 = f_psi(es, ak, pt - h * l1, single)
            l3 = f_psi(es, ak, pt - h * l2, single)
            l4 = f_psi(de_ds[t - 1], lin_a[t - 1], pt - dt * l3, single)
            psi[t - 1] = pt - dt * (l1 + 2.0 * (l2 + l3) + l4) / 6.0 + jump_s[t - 1]
        else:
            raise ValueError(f"unknown integration method: {method}")
    return lam, psi


# --------------------------------------------------------------------------- #
#  Gaussian moments: src/var_bayes/gaussian_moments.py:43-183 (orders used by OU / DW)
# --------------------------------------------------------------------------- #
def gm(m, v, order):
    if order == 2:
        return m ** 2 + v
    if order == 3:
        return m ** 3 + 3 * m * v
    if order == 4:
        return m ** 4 + 6 * (m ** 2) * v + 3 * (v ** 2)
    if order == 6:
        return m ** 6 + 15 * (m ** 4) * v + 45 * (m ** 2) * (v ** 2) + 15 * (v ** 3)
    raise ValueError(order)


def gm_dm(m, v, order):
    if order == 2:
        return 2 * m
    if order == 3:
        return 3 * (m ** 2 + v)
    if order == 4:
        return 4 * (m ** 3 + 3 * m * v)
    if order == 6:
        return 6 * (m ** 5 + 10 * (m ** 3) * v + 15 * m * (v ** 2))
    raise ValueError(order)


def gm_ds(m, v, order):
    if order == 2:
        return np.ones(np.shape(m))
    if order == 3:
        return 3 * m
    if order == 4:
        return 6 * (m ** 2 + v)
    if order == 6:
        return 15 * (m ** 4) + 90 * (m ** 2) * v + 45 * (v ** 2)
    raise ValueError(order)


# --------------------------------------------------------------------------- #
#  Per-model E_sde and gradients
# --------------------------------------------------------------------------- #
def energy_ou(theta, sigma, dt, a, b, m, s, obs_t):
    """src/dynamics/ornstein_uhlenbeck.py:165-232."""
    ex2 = gm(m, s, 2)
    q1 = (theta - a) ** 2
    q2 = a * b
    var_q = ex2 * q1 + 2.0 * m * (theta - a) * b + (b ** 2)
    esde = 0.5 * my_trapz(var_q, dt, obs_t) / sigma
    ef = -theta * m
    edf = -theta * np.ones(m.shape)
    de_dm = (m * (theta - a) ** 2 + theta * b - q2) / sigma
    de_ds = 0.5 * q1 / sigma
    de_dth = my_trapz(ex2 * (theta - a) + m * b, dt, obs_t) / sigma
    de_dsig = -esde / sigma
    return esde, (ef, edf), (de_dm, de_ds, de_dth, de_dsig)


def energy_dw(theta, sigma, dt, a, b, m, s, obs_t):
    """src/dynamics/double_well.py:169-260 (Q7: 8*Ex6 in the energy, 16*Dm6 in its gradient)."""
    c = (4.0 * theta) + a
    c2 = c ** 2
    ex2, ex3, ex4, ex6 = gm(m, s, 2), gm(m, s, 3), gm(m, s, 4), gm(m, s, 6)
    var_q = 8.0 * (ex6 - c * ex4 + b * ex3) + (c2 * ex2) - (2.0 * b * c * m) + (b ** 2)
    esde = 0.5 * my_trapz(var_q, dt, obs_t) / sigma
    ef = 4.0 * (theta * m - ex3)
    edf = 4.0 * (theta - 3.0 * ex2)
    de_dm = 0.5 * (16.0 * gm_dm(m, s, 6) - 8.0 * c * gm_dm(m, s, 4) +
                   8.0 * b * gm_dm(m, s, 3) + c2 * gm_dm(m, s, 2) - 2.0 * b * c) / sigma
    de_ds = 0.5 * (16.0 * gm_ds(m, s, 6) - 8.0 * c * gm_ds(m, s, 4) +
                   8.0 * b * gm_ds(m, s, 3) + c2 * gm_ds(m, s, 2)) / sigma
    de_dth = 4.0 * my_trapz(c * ex2 - 4.0 * ex4 - b * m, dt, obs_t) / sigma
    de_dsig = -esde / sigma
    return esde, (ef, edf), (de_dm, de_ds, de_dth, de_dsig)


def l63_point(theta, at, bt, mt, st, isg):
    """
    Closed-form Gaussian-moment energy of the Lorenz-63 drift at ONE grid point.
    src/dynamics/lorenz_63.py:348-568 (energy_dm_ds).  Returns (Efg, dEsde_dm, dEsde_ds).
    """
    vS, vR, vB = theta
    (A11, A12, A13), (A21, A22, A23), (A31, A32, A33) = at
    b1, b2, b3 = bt
    mx, my, mz = mt
    Sxx, Sxy, Sxz = st[0]
    Syy, Syz = st[1][1], st[1][2]
    Szz = st[2][2]
    # second order
    Exx, Exy, Exz = Sxx + mx ** 2, Sxy + mx * my, Sxz + mx * mz
    Eyy, Eyz, Ezz = Syy + my ** 2, Syz + my * mz, Szz + mz ** 2
    # third order
    Exxy = Sxx * my + 2 * Sxy * mx + (mx ** 2) * my
    Exxz = Sxx * mz + 2 * Sxz * mx + (mx ** 2) * mz
    Exyy = Syy * mx + 2 * Sxy * my + (my ** 2) * mx
    Exzz = Szz * mx + 2 * Sxz * mz + (mz ** 2) * mx
    Exyz = Sxy * mz + Sxz * my + Syz * mx + mx * my * mz
    # fourth order
    Exxyy = Sxx * (my ** 2 + Syy) + Syy * (mx ** 2) + 4.0 * Sxy * mx * my + (mx * my) ** 2 + 2 * (Sxy ** 2)
    Exxzz = Sxx * (mz ** 2 + Szz) + Szz * (mx ** 2) + 4.0 * Sxz * mx * mz + (mx * mz) ** 2 + 2 * (Sxz ** 2)
    # <(f-g)^2> per component (lorenz_63.py:414-436)
    EX = (vS ** 2) * (Eyy + Exx - 2 * Exy) + (A11 ** 2) * Exx + (A12 ** 2) * Eyy + \
         (A13 ** 2) * Ezz + b1 ** 2 + 2 * (A11 * A12 * Exy + A11 * A13 * Exz - b1 * A11 * mx +
                                           A12 * A13 * Eyz - b1 * A12 * my - b1 * A13 * mz +
                                           vS * (A11 * Exy + A12 * Eyy + A13 * Eyz - b1 * my -
                                                 A11 * Exx - A12 * Exy - A13 * Exz + b1 * mx))
    EY = (vR ** 2) * Exx + Eyy + Exxzz + (A21 ** 2) * Exx + (A22 ** 2) * Eyy + \
         (A23 ** 2) * Ezz + b2 ** 2 + 2 * (Exyz - A21 * Exy - A22 * Eyy - A23 * Eyz -
                                           A21 * Exxz - A22 * Exyz - A23 * Exzz +
                                           A21 * A22 * Exy + A21 * A23 * Exz + A22 * A23 * Eyz -
                                           vR * (Exy + Exxz - A21 * Exx - A22 * Exy - A23 * Exz) -
                                           b2 * (vR * mx - my - Exz + A21 * mx + A22 * my + A23 * mz))
    EZ = Exxyy + (vB ** 2) * Ezz + (A31 ** 2) * Exx + (A32 ** 2) * Eyy + (A33 ** 2) * Ezz + \
         b3 ** 2 + 2 * (A31 * Exxy + A32 * Exyy + A33 * Exyz + A31 * A32 * Exy +
                        A31 * A33 * Exz + A32 * A33 * Eyz -
                        vB * (Exyz + A31 * Exz + A32 * Eyz + A33 * Ezz) -
                        b3 * (Exy - vB * mz + A31 * mx + A32 * my + A33 * mz))
    efg = np.array([EX, EY, EZ])
    # d/dm of the moments
    d2x = (2.0 * mx, my, mz)      # dExx_dmx, dExy_dmx, dExz_dmx
    d2y = (2.0 * my, mx, mz)      # dEyy_dmy, dExy_dmy, dEyz_dmy
    d2z = (2.0 * mz, mx, my)      # dEzz_dmz, dExz_dmz, dEyz_dmz
    # lorenz_63.py:497-527
    dmx1 = d2x[0] * (vS ** 2 + A11 ** 2) + 2 * (d2x[1] * (-vS ** 2 + vS * A11 - vS * A12 + A11 * A12) +
                                                d2x[2] * (A11 - vS) * A13 - vS * A11 * d2x[0] + b1 * (vS - A11))
    dmx2 = 2.0 * Exzz + d2x[0] * (vR ** 2 + A21 ** 2) + \
        2 * (d2x[1] * (-vR + vR * A22 - A21 + A21 * A22) + d2x[2] * (vR * A23 + b2 + A21 * A23) +
             Eyz * (1 - A22) - vR * (2.0 * Exz) + vR * A21 * d2x[0] - A21 * (2.0 * Exz) - A23 * Ezz - b2 * (vR + A21))
    dmx3 = 2.0 * Exyy + (A31 ** 2) * d2x[0] + 2 * (d2x[1] * (A31 * A32 - b3) + d2x[2] * (A33 - vB) * A31 +
                                                   Eyz * (A33 - vB) + A31 * (2.0 * Exy) + A32 * Eyy - A31 * b3)
    dmy1 = d2y[0] * (vS ** 2 + A12 ** 2) + 2 * (d2y[1] * (-(vS ** 2) + vS * A11 - vS * A12 + A11 * A12) +
                                                d2y[2] * (vS + A12) * A13 + vS * A12 * d2y[0] - b1 * (vS + A12))
    dmy2 = d2y[0] * (1 + A22 ** 2) + 2 * (d2y[1] * (-vR + vR * A22 - A21 + A21 * A22) + Exz * (1 - A22) -
                                          A22 * d2y[0] + d2y[2] * (A22 * A23 - A23) + b2 * (1 - A22))
    dmy3 = 2.0 * Exxy + (A32 ** 2) * d2y[0] + 2 * (Exz * (A33 - vB) + A31 * Exx + A32 * (2.0 * Exy) +
                                                   d2y[1] * (A31 * A32 - b3) + d2y[2] * (A33 - vB) * A32 - A32 * b3)
    dmz1 = (A13 ** 2) * d2z[0] + 2 * (d2z[2] * (vS + A12) + d2z[1] * (A11 - vS) - b1) * A13
    dmz2 = 2.0 * Exxz + (A23 ** 2) * d2z[0] + 2 * (Exx * (-vR - A21) + d2z[1] * (vR * A23 + b2 + A21 * A23) +
                                                   Exy * (1 - A22) + d2z[2] * (A22 * A23 - A23) -
                                                   A23 * (2.0 * Exz + b2))
    dmz3 = d2z[0] * (vB ** 2 + A33 ** 2) + 2 * ((A33 - vB) * (Exy + d2z[1] * A31 + d2z[2] * A32 - b3) -
                                                vB * A33 * d2z[0])
    de_dm = 0.5 * np.array([[dmx1, dmx2, dmx3], [dmy1, dmy2, dmy3], [dmz1, dmz2, dmz3]]).dot(isg)
    iSx, iSy, iSz = isg
    # lorenz_63.py:535-561 (the unit derivatives dE??_dS?? = 1 are folded in)
    dSxx = iSx * ((vS - A11) ** 2) + iSy * (Ezz + ((vR + A21) ** 2) - 2 * mz * (vR + A21)) + \
        iSz * (Eyy + (A31 ** 2) + 2 * A31 * my)
    dSxy = iSx * 2 * (vS * A11 - vS ** 2 - vS * A12 + A11 * A12) + \
        iSy * 2 * ((vR * A22 - vR - A21 + A21 * A22) + mz * (1 - A22)) + \
        iSz * (4.0 * Exy + 2 * (mz * (A33 - vB) + A31 * (2.0 * mx) + A32 * (2.0 * my) + (A31 * A32 - b3)))
    dSxz = iSx * 2 * (A11 - vS) * A13 + \
        iSy * (4.0 * Exz + 2 * ((vR * A23 + b2 + A21 * A23) + my * (1 - A22) - (2.0 * mx) * (vR + A21) -
                                A23 * (2.0 * mz))) + \
        iSz * 2 * ((A33 - vB) * A31 + my * (A33 - vB))
    dSyy = iSx * ((vS + A12) ** 2) + iSy * ((1 - A22) ** 2) + iSz * (Exx + (A32 ** 2) + 2 * A32 * mx)
    dSyz = iSx * 2 * (vS + A12) * A13 + iSy * 2 * (mx * (1 - A22) + (A22 - 1) * A23) + \
        iSz * 2 * (mx * (A33 - vB) + (A33 - vB) * A32)
    dSzz = iSx * (A13 ** 2) + iSy * (Exx + (A23 ** 2) - 2 * A23 * mx) + iSz * ((vB - A33) ** 2)
    de_ds = 0.5 * np.array([[dSxx, dSxy, dSxz], [dSxy, dSyy, dSyz], [dSxz, dSyz, dSzz]])
    return efg, de_dm, de_ds


def l63_drift_theta(theta, at, bt, mt, st):
    """src/dynamics/lorenz_63.py:572-633 (<(f-g)' df/dtheta>; unused downstream)."""
    vS, vR, vB = theta
    (A11, A12, A13), (A21, A22, A23), (A31, A32, A33) = at
    b1, b2, b3 = bt
    mx, my, mz = mt
    Sxx, Sxy, Sxz = st[0]
    Syy, Syz = st[1][1], st[1][2]
    Szz = st[2][2]
    Exx, Exy, Eyy = Sxx + mx ** 2, Sxy + mx * my, Syy + my ** 2
    Exz, Ezz, Eyz = Sxz + mx * mz, Szz + mz ** 2, Syz + my * mz
    Exxz = Sxx * mz + 2 * Sxz * mx + (mx ** 2) * mz
    Exyz = Sxy * mz + Sxz * my + Syz * mx + mx * my * mz
    v1 = Eyy * (vS + A12) + Exx * (vS - A11) + Exy * (A11 - 2 * vS - A12) + A13 * (Eyz - Exz) + b1 * (mx - my)
    v2 = vR * Exx - Exy - Exxz + A21 * Exx + A22 * Exy + A23 * Exz - b2 * mx
    v3 = -Exyz + vB * Ezz - A31 * Exz - A32 * Eyz - A33 * Ezz + b3 * mz
    return np.array([v1, v2, v3])


def energy_l63(theta, inv_sigma, dt, lin_a, off_b, m, s, obs_t):
    """src/dynamics/lorenz_63.py:237-346."""
    n = m.shape[0]
    isg = np.diag(inv_sigma)
    e_t = np.zeros(n)
    ef, edf = np.zeros((n, 3)), np.zeros((n, 3, 3))
    de_dm, de_ds = np.zeros((n, 3)), np.zeros((n, 3, 3))
    dth, dsg = np.zeros((n, 3)), np.zeros((n, 3))
    vS, vR, vB = theta
    for t in range(n):
        mt, st = m[t], s[t]
        efg, de_dm[t], de_ds[t] = l63_point(theta, lin_a[t], off_b[t], mt, st, isg)
        e_t[t] = 0.5 * isg.dot(efg)
        ef[t] = np.array([vS * (mt[1] - mt[0]),
                          vR * mt[0] - mt[1] - st[2, 0] - mt[0] * mt[2],
                          st[1, 0] + mt[0] * mt[1] - vB * mt[2]])
        edf[t] = np.array([[-vS, vS, 0], [vR - mt[2], -1, -mt[0]], [mt[1], mt[0], -vB]])
        dth[t] = l63_drift_theta(theta, lin_a[t], off_b[t], mt, st)
        dsg[t] = efg
    esde = my_trapz(e_t, dt, obs_t)
    de_dth = isg * my_trapz(dth, dt, obs_t)
    de_dsig = -0.5 * inv_sigma.dot(np.diag(my_trapz(dsg, dt, obs_t))).dot(inv_sigma)
    return esde, (ef, edf), (de_dm, de_ds, de_dth, de_dsig)


def l96_drift(x, theta):
    """
    src/dynamics/lorenz_96.py:86-101 with :28-32.  `np.roll` WITHOUT axis: on the (M, D)
    sigma-point matrix the shift runs over the FLATTENED array (Q1).
    """
    return (np.roll(x, -1) - np.roll(x, +2)) * np.roll(x, +1) - x + theta


def l96_mean_drift(mt, st, theta):
    """Lorenz96.E96_drift, lorenz_96.py:440-462."""
    idx = np.arange(mt.size)
    f1, b1, b2 = np.roll(idx, -1), np.roll(idx, +1), np.roll(idx, +2)
    cxx = st[f1, b1] - st[b2, b1]
    return cxx + (np.roll(mt, -1) - np.roll(mt, +2)) * np.roll(mt, +1) - mt + theta


def l96_mean_jacobian(x):
    """E96_drift_dx, lorenz_96.py:35-83: 4 non-zeros per row, evaluated at the mean."""
    d = x.size
    idx = np.arange(d)
    f1i, b1i, b2i = np.roll(idx, -1), np.roll(idx, +1), np.roll(idx, +2)
    f1x, b1x, b2x = np.roll(x, -1), np.roll(x, +1), np.roll(x, +2)
    jac = np.zeros((d, d))
    for k in range(d):
        row = np.zeros(d)
        row[k] = -1
        row[f1i[k]] = b1x[k]
        row[b2i[k]] = -b1x[k]
        row[b1i[k]] = f1x[k] - b2x[k]
        jac[k] = row
    return jac


def grad_esde_dm_ds(x, fun, mt, st, at, bt, diag_inv_sigma):
    """src/var_bayes/variational.py:339-400: per-sigma-point gradient terms, (M, D + D*D)."""
    n, d = x.shape
    dst = np.zeros((n, d * d))
    x_mat = (fun(x) + x.dot(at.T) - np.tile(bt, (n, 1))) ** 2
    var = diag_inv_sigma.dot(x_mat.T)
    dmt = np.linalg.solve(st, (np.tile(var, (d, 1)) * x.T)).T
    inv_st, _ = chol_inv(st)
    for k in range(n):
        zt = x[k] - mt
        dst[k] = var[k] * np.linalg.solve(st, np.outer(zt, zt)).dot(inv_st).ravel()
    return np.concatenate((0.5 * dmt, 0.5 * dst), axis=1)


def _l96_point_lean(theta, isg, at, bt, mt, st):
    """
    Lean evaluation of one grid point (same maths as the two ut_approx calls, SURVEY.md s.8a
    "Algebra the kernels may exploit").  Valid on the normal Cholesky branch only; a
    non-positive-definite S_t raises LinAlgError exactly as the reference does (its fallback
    factor at utilities.py:279 is followed by chol_inv(S_t) at variational.py:380, which raises).
    """
    d = mt.size
    kappa = 1.05 * d
    c = d + kappa
    low = np.linalg.cholesky(c * st)
    chi = np.concatenate((mt[np.newaxis, :], mt + low.T, mt - low.T))
    resid = (l96_drift(chi, theta) + chi.dot(at.T) - bt) ** 2
    v = resid.dot(isg)                                  # (M,)
    w0, w = kappa / c, 1.0 / (2.0 * c)
    m_bar = w0 * resid[0] + w * np.sum(resid[1:], axis=0)
    e_t = 0.5 * isg.dot(m_bar)
    linv = np.linalg.solve(low, np.eye(d))
    delta = w * (v[1:d + 1] - v[d + 1:])
    e_sum = w * (v[1:d + 1] + v[d + 1:])
    de_dm = 0.5 * c * linv.T.dot(delta)
    de_ds = 0.5 * c * (linv.T * (0.5 * c * e_sum - e_t)).dot(linv)
    return m_bar, e_t, de_dm, de_ds


def energy_l96(theta, inv_sigma, dt, lin_a, off_b, m, s, obs_t, faithful=True):
    """src/dynamics/lorenz_96.py:316-438."""
    n, d = m.shape
    isg = np.diag(inv_sigma)
    e_t = np.zeros(n)
    ef, edf = np.zeros((n, d)), np.zeros((n, d, d))
    de_dm, de_ds = np.zeros((n, d)), np.zeros((n, d, d))
    dth, dsg = np.zeros((n, d)), np.zeros((n, d))
    eye = np.eye(d)

    def fun_1(xt, at, bt):
        return (l96_drift(xt, theta) + xt.dot(at.T) - np.tile(bt, (xt.shape[0], 1))) ** 2

    def fun_2(xt):
        return l96_drift(xt, theta)

    for t in range(n):
        at, bt, mt, st = lin_a[t], off_b[t], m[t], s[t]
        ef[t] = l96_mean_drift(mt, st, theta)
        edf[t] = l96_mean_jacobian(mt)
        if faithful:
            m_bar, _ = ut_approx(fun_1, mt, st, at, bt)
            e_t[t] = 0.5 * isg.dot(m_bar.T)
            dms, _ = ut_approx(grad_esde_dm_ds, mt, st, fun_2, mt, st, at, bt, isg)
            de_dm[t] = dms[:d] - e_t[t] * np.linalg.solve(st, mt)
            de_ds[t] = 0.5 * (dms[d:].reshape(d, d) - e_t[t] * np.linalg.solve(st, eye))
        else:
            m_bar, e_t[t], de_dm[t], de_ds[t] = _l96_point_lean(theta, isg, at, bt, mt, st)
        dth[t] = ef[t] + mt.dot(at.T) - bt
        dsg[t] = m_bar
    esde = my_trapz(e_t, dt, obs_t)
    de_dth = isg * my_trapz(dth, dt, obs_t)
    de_dsig = -0.5 * inv_sigma.dot(np.diag(my_trapz(dsg, dt, obs_t))).dot(inv_sigma)
    return esde, (ef, edf), (de_dm, de_ds, de_dth, de_dsig)


def model_energy(p, lin_a, off_b, m, s, faithful=True):
    """Dispatch on the model name; obs_t is passed as a python list like stochastic_process.py:175."""
    obs_t = list(p.obs_t)
    if p.model == "OU":
        return energy_ou(p.theta, p.sigma, p.dt, lin_a, off_b, m, s, obs_t)
    if p.model == "DW":
        return energy_dw(p.theta, p.sigma, p.dt, lin_a, off_b, m, s, obs_t)
    if p.model == "L63":
        return energy_l63(p.theta, p.inverse_sigma, p.dt, lin_a, off_b, m, s, obs_t)
    if p.model == "L96":
        return energy_l96(p.theta, p.inverse_sigma, p.dt, lin_a, off_b, m, s, obs_t, faithful=faithful)
    raise ValueError(p.model)


# --------------------------------------------------------------------------- #
#  Observation energy: src/var_bayes/gaussian_like.py:69-243, likelihood.py:13-48
# --------------------------------------------------------------------------- #
def _obs_operator(p):
    if p.obs_h is not None:
        return np.asarray(p.obs_h)
    return np.asarray(1) if p.single_dim else np.eye(np.asarray(p.obs_y)[0].size)


def eobs(p, m, s):
    obs_y, obs_t, r = p.obs_y, p.obs_t, p.obs_noise
    if p.single_dim:                                              # gaussian_like.py:69-96
        ex2 = (m[obs_t] ** 2) + s[obs_t]
        return 0.5 * np.sum((obs_y ** 2) - 2.0 * obs_y * m[obs_t] + ex2) / r + \
            0.5 * obs_t.size * (LOG2PI + np.log(r))
    dim_m, dim_o = obs_y.shape                                    # gaussian_like.py:98-153
    w = (obs_y - m[obs_t]).dot(_obs_operator(p))
    inv_r, inv_c = chol_inv(r)
    z = w.dot(inv_c.T)
    s_diag = np.diagonal(s, axis1=1, axis2=2)
    acc = 0.0
    for n in range(dim_m):
        # Q4: the covariance diagonal is indexed by the observation COUNTER n, not by obs_t[n].
        acc += np.inner(z[n], z[n]) + np.inner(inv_r.diagonal(), s_diag[n])
    return 0.5 * (acc + dim_m * (dim_o * LOG2PI + log_det(r)))


def eobs_gradients(p, m, s):
    obs_y, obs_t, r = p.obs_y, p.obs_t, p.obs_noise
    h = _obs_operator(p)
    if p.single_dim:                                              # gaussian_like.py:155-198
        n = m.shape[0]
        jm, js = np.zeros(n), np.zeros(n)
        jm[obs_t] = -(obs_y - h * m[obs_t]) / r
        js[obs_t] = 0.5 / r
        return jm, js
    n, d = m.shape                                                # gaussian_like.py:200-243
    w = (obs_y - m[obs_t]).dot(h)
    inv_r, _ = chol_inv(r)
    jm, js = np.zeros((n, d)), np.zeros((n, d, d))
    for k, tn in enumerate(obs_t):
        jm[tn] = -h.T.dot(inv_r).dot(w[k])
        js[tn] = 0.5 * h.T.dot(inv_r).dot(h)
    return jm, js


# --------------------------------------------------------------------------- #
#  KL at t = 0: src/var_bayes/prior_kl0.py:46-92
# --------------------------------------------------------------------------- #
def kl0(p):
    m0, s0, mu0, tau0 = p.m0, p.s0, np.asarray(p.mu0), np.asarray(p.tau0)
    z0 = m0 - mu0
    if p.single_dim:
        return -np.log(s0) - 0.5 * (1.0 - np.log(tau0)) + 0.5 / tau0 * (z0 ** 2 + s0)
    inv_tau0, _ = chol_inv(tau0)
    inv_s0, _ = chol_inv(s0)
    # Q5: z0.T.dot(z0) is a SCALAR broadcast over the matrix; log_det of a non-symmetric product.
    return 0.5 * (log_det(tau0.dot(inv_s0)) + np.sum(np.diag(inv_tau0.dot(z0.T.dot(z0) + s0 - tau0))))


# --------------------------------------------------------------------------- #
#  The objective: src/var_bayes/variational.py:141-334
# --------------------------------------------------------------------------- #
def free_energy(p, x, faithful=True):
    """VarGP.free_energy, variational.py:141-200.  Returns (F, state dict)."""
    lin_a, off_b = p.split(x)
    single = p.single_dim
    mt, st = solve_fwd(p.method, p.dt, single, lin_a, off_b, p.m0, p.s0, p.sigma)
    e_obs = eobs(p, mt, st)
    e_sde, (efx, edf), (de_dm, de_ds, de_dth, de_dsig) = model_energy(p, lin_a, off_b, mt, st, faithful)
    jm, js = eobs_gradients(p, mt, st)
    lam, psi = solve_bwd(p.method, p.dt, single, lin_a, de_dm, de_ds, jm, js)
    e0 = kl0(p)
    state = dict(mt=mt, st=st, Efx=efx, Edf=edf, lamt=lam, psit=psi, Esde=e_sde, Eobs=e_obs, E0=e0,
                 dEsde_dm=de_dm, dEsde_ds=de_ds, dEobs_dm=jm, dEobs_ds=js,
                 dEsde_dth=de_dth, dEsde_dSig=de_dsig)
    return np.array(e0 + e_sde + e_obs).item(), state


def gradient(p, x, state):
    """VarGP.gradient with the cached state, variational.py:202-334."""
    lin_a, off_b = p.split(x)
    n, d = p.n_pts, p.dim_d
    mt, st, lam, psi, efx, edf = (state[k] for k in ("mt", "st", "lamt", "psit", "Efx", "Edf"))
    inv_sigma = p.inverse_sigma
    if p.single_dim:
        gla, glb = np.zeros(n), np.zeros(n)
    else:
        gla, glb = np.zeros((n, d, d)), np.zeros((n, d))
    for k in range(n):
        ak, sk, mk, lk = lin_a[k], st[k], mt[k], lam[k]
        if p.single_dim:
            db = inv_sigma * (-efx[k] - (ak * mk) + off_b[k])                   # :330
            da = inv_sigma * (edf[k] + ak) * sk - (db * mk)                     # :318
            gla[k] = da - (lk * mk) - (2.0 * psi[k] * sk)                       # :306
        else:
            db = inv_sigma.dot(-efx[k] - ak.dot(mk) + off_b[k])                 # :332
            da = inv_sigma.dot(edf[k] + ak).dot(sk) - np.outer(db, mk)          # :320
            gla[k] = da - np.outer(lk, mk) - 2.0 * psi[k].dot(sk)               # :308
        glb[k] = db + lk
    return np.concatenate(((p.dt * gla).flatten(), (p.dt * glb).flatten()))


def sweep(p, x, faithful=True):
    """One fwd+bwd sweep == VarGP.gradient(x, eval_fun=True): returns (F, grad, state)."""
    f, state = free_energy(p, x, faithful=faithful)
    return f, gradient(p, x, state), state
