"""
Host-side mirror of the reference's `stochastic_process` plugin surface (src/dynamics/*.py).

Each model keeps the reference's constructor, properties (`sigma`, `inverse_sigma`, `theta`, `single_dim`,
`time_step`, `time_window`, `sample_path`, `rng`), `make_trajectory`, `collect_obs` and

    energy(linear_a, offset_b, m, s, obs_t) -> (Esde, (Ef, Edf), (dEsde_dm, dEsde_ds, dEsde_dth, dEsde_dSig))

but `energy` runs on the GPU through libvgpa_hip.so (vgpa_energy).  The hyper-parameter gradients
(`dEsde_dth`, `dEsde_dSig`; unused by VarGP) are returned as None.

The synthetic-data generators (`make_trajectory`, `collect_obs`) are one-off host-side input builders
(numpy); they consume the random stream in exactly the reference's order, so that a given seed yields the
same sample path and observations (stochastic_process.py:130-230, lorenz_96.py:249-314,
lorenz_63.py:174-234, ornstein_uhlenbeck.py:122-163, double_well.py:118-167).

Deliberate deviation: a scalar `sigma` is accepted by the n-D models (the reference turns it into a 0-d array
and then raises "Wrong input dimensions: 0", which makes its own shipped JSON files unusable).
"""
import numpy as np
from numpy.random import default_rng, SeedSequence
from scipy.linalg import cholesky as _upper_cholesky, LinAlgError

from ._lib import Context


class StochasticProcess(object):
    """Base class: random stream, sample path, time window (src/dynamics/stochastic_process.py:5-232)."""

    def __init__(self, r_seed=None, single_dim: bool = True) -> None:
        self.rand_g = default_rng(SeedSequence(r_seed)) if r_seed else default_rng()
        self.single_dimension = single_dim
        self.xt = None
        self.tk = None
        self._energy_ctx = {}
        self.device = 0

    # -- accessors ---------------------------------------------------------------------------
    @property
    def single_dim(self) -> bool:
        return self.single_dimension

    @property
    def sample_path(self) -> np.ndarray:
        if self.xt is None:
            raise NotImplementedError(f" {self.__class__.__name__}: Sample path has not been created.")
        return self.xt

    @sample_path.setter
    def sample_path(self, new_value: np.ndarray) -> None:
        self.xt = new_value

    @property
    def time_window(self) -> np.ndarray:
        if self.tk is None:
            raise NotImplementedError(f" {self.__class__.__name__}: Time window has not been created yet.")
        return self.tk

    @time_window.setter
    def time_window(self, new_value: np.ndarray) -> None:
        self.tk = new_value

    @property
    def time_step(self):
        if self.tk is None:
            raise NotImplementedError(f" {self.__class__.__name__}: Time window has not been created yet.")
        return np.abs(self.tk[1] - self.tk[0])

    @property
    def rng(self):
        return self.rand_g

    # -- observations ------------------------------------------------------------------------
    def collect_obs(self, n_obs: int, rn, h_mask=None):
        """Equidistant noisy observations of the sample path (stochastic_process.py:130-230)."""
        if self.tk is None or self.xt is None:
            raise NotImplementedError(f" {self.__class__.__name__}:"
                                      f" Sample path (or time window) have not been created.")
        rn = np.asarray(rn)
        step = np.diff(self.tk)[0]
        if n_obs > int(1.0 / step):
            raise ValueError(f" {self.__class__.__name__}:"
                             f" Observation density exceeds the number of samples.")
        n_total = int(np.floor(np.abs(self.tk[0] - self.tk[-1]) * n_obs))
        grid = np.linspace(0, self.tk.size, n_total + 2, dtype=int)
        obs_t = sorted(np.unique(grid[1:-1]))
        obs_y = np.take(self.xt, obs_t, axis=0)
        if h_mask:
            obs_y = obs_y[:, h_mask]
        dim_o = 1 if obs_y.ndim == 1 else obs_y.shape[-1]
        if dim_o == 1:
            obs_noise = rn
            obs_y += np.sqrt(obs_noise) * self.rand_g.standard_normal(n_total)
        else:
            obs_noise = np.diag(rn) if rn.ndim == 1 else rn * np.eye(dim_o)
            obs_y += np.sqrt(obs_noise).dot(self.rand_g.standard_normal((dim_o, n_total))).T
        return obs_t, obs_y, obs_noise

    # -- device plumbing ---------------------------------------------------------------------
    _model_id = "NONE"

    def _theta_vec(self):
        return np.atleast_1d(np.asarray(self.theta, dtype=float))

    def _ctx_for(self, n_pts, dim_d):
        key = (n_pts, dim_d, self.device)
        ctx = self._energy_ctx.get(key)
        if ctx is None:
            self._energy_ctx.clear()
            sigma = np.array([[self.sigma]], dtype=float) if self.single_dim else np.asarray(self.sigma, dtype=float)
            ctx = Context(self._model_id, "euler", dim_d, n_pts, float(self.time_step), sigma=sigma,
                          theta=self._theta_vec(), device=self.device)
            self._energy_ctx[key] = ctx
        return ctx

    def _invalidate(self):
        for c in self._energy_ctx.values():
            c.close()
        self._energy_ctx.clear()

    def energy(self, linear_a, offset_b, m, s, obs_t):
        """E_sde and related quantities on the GPU.  `obs_t` is accepted for signature parity: the piecewise
        trapezoid of the reference (utilities.py:144-201) equals one global trapezoid."""
        m = np.asarray(m, dtype=float)
        if self.single_dim:
            n = m.size
            ctx = self._ctx_for(n, 1)
            esde, ef, edf, dm, ds, dth, dsg = ctx.energy(np.asarray(linear_a, dtype=float).reshape(n, 1, 1),
                                                         np.asarray(offset_b, dtype=float).reshape(n, 1),
                                                         m.reshape(n, 1), np.asarray(s, dtype=float).reshape(n, 1, 1),
                                                         want_hyper=True)
            return esde, (ef.reshape(n), edf.reshape(n)), (dm.reshape(n), ds.reshape(n), dth, dsg)
        n, d = m.shape
        ctx = self._ctx_for(n, d)
        esde, ef, edf, dm, ds, dth, dsg = ctx.energy(linear_a, offset_b, m, s, want_hyper=True)
        return esde, (ef, edf), (dm, ds, dth, dsg)


def _as_noise_matrix(cls_name, sigma, dim_d):
    sigma = np.asarray(sigma, dtype=float)
    if sigma.ndim == 0:
        mat = float(sigma) * np.eye(dim_d)
    elif sigma.ndim == 1:
        mat = np.diag(sigma)
    elif sigma.ndim == 2:
        mat = sigma
    else:
        raise ValueError(f" {cls_name}: Wrong input dimensions: {sigma.ndim}")
    if mat.shape != (dim_d, dim_d):
        raise ValueError(f" {cls_name}: Wrong matrix dimensions: {mat.shape}")
    if not np.all(np.linalg.eigvals(mat) > 0.0):
        raise RuntimeError(f" {cls_name}: Noise matrix {mat} is not positive definite.")
    return mat


def _spd_inverse(x):
    """chol_inv of the reference (utilities.py:203-237): (L^-1)^T (L^-1)."""
    c_inv = np.linalg.solve(np.linalg.cholesky(x), np.eye(x.shape[0]))
    return c_inv.T.dot(c_inv)


def _noise_increments(rng, sigma, dt, dim_d, dim_t):
    """chol(Sigma*dt) . N(0,1)^{D x T}, transposed (lorenz_96.py:289-301; scipy's upper factor)."""
    try:
        root = _upper_cholesky(sigma * dt)
    except LinAlgError:
        print(" Warning : The input matrix was not positive definite."
              " The diagonal elements will be used instead.")
        root = np.sqrt(np.eye(dim_d) * sigma * dt)
    return root.dot(rng.standard_normal((dim_d, dim_t))).T


class _ScalarDiffusion(StochasticProcess):
    """Common part of the two 1-D models."""

    def __init__(self, sigma: float, theta: float, r_seed=None) -> None:
        super().__init__(r_seed, single_dim=True)
        if sigma <= 0.0:
            raise ValueError(f" {self.__class__.__name__}: The diffusion noise value: {sigma},"
                             f" should be strictly positive.")
        self._sigma = sigma
        self._check_theta(theta)
        self._theta = theta
        self.sig_inv = 1.0 / sigma

    def _check_theta(self, theta):
        pass

    @property
    def theta(self):
        return self._theta

    @theta.setter
    def theta(self, new_value) -> None:
        self._check_theta(new_value)
        self._theta = new_value
        self._invalidate()

    @property
    def sigma(self):
        return self._sigma

    @sigma.setter
    def sigma(self, new_value) -> None:
        if new_value <= 0.0:
            raise ValueError(f"{self.__class__.__name__}: The sigma value {new_value}, should be strictly positive.")
        self._sigma = new_value
        self.sig_inv = 1.0 / new_value
        self._invalidate()

    @property
    def inverse_sigma(self):
        return self.sig_inv


class OrnsteinUhlenbeck(_ScalarDiffusion):
    """dx = theta (mu - x) dt + sqrt(sigma) dW (src/dynamics/ornstein_uhlenbeck.py)."""

    _model_id = "OU"

    def __init__(self, sigma: float, theta: float, r_seed=None) -> None:
        super().__init__(sigma, theta, r_seed)
        print(" Creating Ornstein-Uhlenbeck process.")

    def _check_theta(self, theta):
        if theta <= 0.0:
            raise ValueError(f" {self.__class__.__name__}: The drift parameter: {theta},"
                             f" should be strictly positive.")

    def make_trajectory(self, t0: float, tf: float, dt: float = 0.01, mu: float = 0.0) -> None:
        tk = np.arange(t0, tf + dt, dt)
        x = np.zeros(tk.size)
        x[0] = mu
        noise = np.sqrt(self._sigma * dt) * self.rng.standard_normal(tk.size)
        for t in range(1, tk.size):
            x[t] = x[t - 1] + self._theta * (mu - x[t - 1]) * dt + noise[t]
        self.sample_path, self.time_window = x, tk


class DoubleWell(_ScalarDiffusion):
    """dx = 4x(theta - x^2) dt + sqrt(sigma) dW (src/dynamics/double_well.py)."""

    _model_id = "DW"

    def __init__(self, sigma: float, theta: float, r_seed=None) -> None:
        super().__init__(sigma, theta, r_seed)
        print(" Creating Double-Well process.")

    def make_trajectory(self, t0: float, tf: float, dt: float = 0.01) -> None:
        tk = np.arange(t0, tf + dt, dt)
        x = np.zeros(tk.size)
        x[0] = +self._theta if self.rng.random() > 0.5 else -self._theta
        x[0] += np.sqrt(0.5 * self._sigma * dt) * self.rng.standard_normal()
        noise = np.sqrt(self._sigma * dt) * self.rng.standard_normal(tk.size)
        for t in range(1, tk.size):
            x[t] = x[t - 1] + 4.0 * x[t - 1] * (self._theta - x[t - 1] ** 2) * dt + noise[t]
        self.sample_path, self.time_window = x, tk


class _VectorDiffusion(StochasticProcess):
    """Common part of the two Lorenz models."""

    def __init__(self, sigma, theta, r_seed, dim_d):
        super().__init__(r_seed, single_dim=False)
        self.dim_d = dim_d
        self._sigma = _as_noise_matrix(self.__class__.__name__, sigma, dim_d)
        self.sig_inv = _spd_inverse(self._sigma)
        self._theta = np.asarray(theta)

    @property
    def theta(self):
        return self._theta

    @theta.setter
    def theta(self, new_value) -> None:
        self._theta = new_value
        self._invalidate()

    @property
    def sigma(self):
        return self._sigma

    @sigma.setter
    def sigma(self, new_value) -> None:
        self._sigma = _as_noise_matrix(self.__class__.__name__, new_value, self.dim_d)
        self.sig_inv = _spd_inverse(self._sigma)
        self._invalidate()

    @property
    def inverse_sigma(self):
        return self.sig_inv

    def _drift(self, x):
        raise NotImplementedError

    def _burn_in_start(self):
        raise NotImplementedError

    def make_trajectory(self, t0: float, tf: float, dt: float = 0.01) -> None:
        tk = np.arange(t0, tf + dt, dt)
        x0 = self._burn_in_start()
        for _ in range(5000):                       # burn-in with step 1e-3
            x0 = x0 + self._drift(x0) * 1.0e-3
        x = np.zeros((tk.size, self.dim_d))
        x[0] = x0
        noise = _noise_increments(self.rng, self._sigma, dt, self.dim_d, tk.size)
        for t in range(1, tk.size):
            x[t] = x[t - 1] + self._drift(x[t - 1]) * dt + noise[t]
        self.sample_path, self.time_window = x, tk


class Lorenz63(_VectorDiffusion):
    """Stochastic Lorenz-63 (src/dynamics/lorenz_63.py)."""

    _model_id = "L63"

    def __init__(self, sigma, theta, r_seed=None) -> None:
        super().__init__(sigma, theta, r_seed, 3)
        print(" Creating Lorenz-63 process.")

    def _drift(self, state):
        x, y, z = state
        s, r, b = self._theta
        return np.array([s * (y - x), (r - z) * x - y, x * y - b * z])

    def _burn_in_start(self):
        return np.ones(3)


class Lorenz96(_VectorDiffusion):
    """Stochastic Lorenz-96 (src/dynamics/lorenz_96.py); `dim_d` defaults to 40 like the reference."""

    _model_id = "L96"

    def __init__(self, sigma, theta: float, r_seed=None, dim_d: int = 40) -> None:
        if dim_d < 10:
            raise ValueError(f" {self.__class__.__name__}: Insufficient state vector dimensions: {dim_d}")
        super().__init__(sigma, theta, r_seed, dim_d)
        print(f" Creating Lorenz-96 (D={dim_d}) process.")

    def _drift(self, x):
        # circular on a single state vector (lorenz_96.py:86-101)
        return (np.roll(x, -1) - np.roll(x, +2)) * np.roll(x, +1) - x + self._theta

    def _burn_in_start(self):
        x0 = self._theta * np.ones(self.dim_d)
        x0[int(self.dim_d / 2.0)] += 1.0e-3
        return x0


# Registry with the reference's keys (src/var_bayes/simulation.py:20-21).
dynamical_systems = {"DW": DoubleWell, "OU": OrnsteinUhlenbeck, "L63": Lorenz63, "L96": Lorenz96}
