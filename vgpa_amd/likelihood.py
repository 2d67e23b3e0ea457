"""
Observation likelihood and the KL term at t=0: host-side mirrors of
src/var_bayes/likelihood.py, src/var_bayes/gaussian_like.py and src/var_bayes/prior_kl0.py.

GaussianLikelihood evaluates E_obs and the jump arrays on the GPU (vgpa_obs_energy).  PriorKL0 is a
constant of the optimisation (m0, s0 are fixed, variational.py:22-26) and is evaluated once on the host.
"""
import numpy as np

from ._lib import Context


class Likelihood(object):
    """Stores observation values / times / noise / operator (src/var_bayes/likelihood.py:13-48)."""

    def __init__(self, values, times, noise, operator=None) -> None:
        self.obs_t = np.asarray(times)
        self.obs_v = np.asarray(values)
        self.obs_n = np.asarray(noise)
        self.default_operator = operator is None
        if operator is None:
            y0 = self.obs_v[0]
            self.obs_h = np.asarray(1) if y0.ndim == 0 else np.eye(y0.size)
        else:
            self.obs_h = np.asarray(operator)

    @property
    def values(self):
        return self.obs_v

    @property
    def times(self):
        return self.obs_t

    @property
    def noise(self):
        return self.obs_n

    @noise.setter
    def noise(self, new_value):
        self.obs_n = new_value
        self._drop_contexts()

    def _drop_contexts(self):
        """Device contexts bake the noise in; subclasses that hold some close them here."""

    @property
    def operator(self):
        return self.obs_h


class GaussianLikelihood(Likelihood):
    """Gaussian likelihood (src/var_bayes/gaussian_like.py:69-243); quirk Q4 of the n-D energy is kept."""

    LOG2PI = np.log(2.0 * np.pi)

    def __init__(self, values, times, noise, operator, single_dim: bool = True) -> None:
        super().__init__(values, times, noise, operator)
        self.single_dim = single_dim
        self._ctx = {}
        self.device = 0

    def _drop_contexts(self):
        for ctx in getattr(self, "_ctx", {}).values():
            ctx.close()
        self._ctx = {}

    def _context(self, n_pts, dim_d):
        key = (n_pts, dim_d, self.device)
        ctx = self._ctx.get(key)
        if ctx is None:
            self._drop_contexts()
            obs_h = None if self.default_operator else np.asarray(self.operator, dtype=float).reshape(dim_d, dim_d)
            ctx = Context("NONE", "euler", dim_d, n_pts, 1.0, sigma=np.eye(dim_d), obs_t=self.times,
                          obs_y=self.values, obs_noise=np.asarray(self.noise, dtype=float).reshape(dim_d, dim_d),
                          obs_h=obs_h, device=self.device)
            self._ctx[key] = ctx
        return ctx

    def _run(self, m, s, want_jumps):
        m = np.asarray(m, dtype=float)
        if self.single_dim:
            n = m.size
            ctx = self._context(n, 1)
            if s is None:
                s = np.zeros(n)
            out = ctx.obs_energy(m.reshape(n, 1), np.asarray(s, dtype=float).reshape(n, 1, 1), want_jumps)
            if not want_jumps:
                return out
            return out[0], out[1].reshape(n), out[2].reshape(n)
        n, d = m.shape
        ctx = self._context(n, d)
        if s is None:
            s = np.zeros((n, d, d))
        return ctx.obs_energy(m, s, want_jumps)

    def __call__(self, m, s):
        return self._run(m, s, False)

    def gradients(self, m, s=None):
        """(dEobs_dm, dEobs_ds, dEobs_dr) (gaussian_like.py:154-243).  The jump terms come from the GPU; dEobs_dr --
        which no caller consumes -- is the O(M) host expression in 1-D and, as in the reference, an all-zero
        (dim_n, dim_o, dim_o) array in n-D (gaussian_like.py:230 allocates it and never fills it)."""
        _, jm, js = self._run(m, s, True)
        obs_t = np.asarray(self.times)
        if self.single_dim:
            mo, so = np.asarray(m, dtype=float)[obs_t], np.asarray(s, dtype=float)[obs_t]
            y = np.asarray(self.values, dtype=float)
            d_r = np.zeros(np.asarray(m).shape[0])
            d_r[obs_t] = -0.5 * ((y ** 2) - 2.0 * y * mo + ((mo ** 2) + so) + 1.0) / self.noise
        else:
            dim_o = np.asarray(self.values).shape[0]
            d_r = np.zeros((np.asarray(m).shape[0], dim_o, dim_o))
        return jm, js, d_r


class PriorKL0(object):
    """KL(q0 || p0) (src/var_bayes/prior_kl0.py:30-92) and its gradients (:94-175); host side: constant in x."""

    def __init__(self, mu0, tau0, single_dim: bool = True) -> None:
        self.mu0 = np.asarray(mu0)
        self.tau0 = np.asarray(tau0)
        self.single_dim = single_dim

    def __call__(self, m0, s0):
        z0 = m0 - self.mu0
        if self.single_dim:
            return -np.log(s0) - 0.5 * (1.0 - np.log(self.tau0)) + 0.5 / self.tau0 * (z0 ** 2 + s0)

        def spd_inv(x):
            c_inv = np.linalg.solve(np.linalg.cholesky(x), np.eye(x.shape[0]))
            return c_inv.T.dot(c_inv)

        def chol_logdet(x):        # Cholesky of the lower triangle, also for a non-symmetric product (Q5)
            return 2.0 * np.sum(np.log(np.linalg.cholesky(x).diagonal()))

        inv_tau0, inv_s0 = spd_inv(self.tau0), spd_inv(np.asarray(s0))
        # Q5: z0.T.dot(z0) is a scalar that numpy broadcasts over the whole matrix.
        return 0.5 * (chol_logdet(self.tau0.dot(inv_s0)) +
                      np.sum(np.diag(inv_tau0.dot(z0.T.dot(z0) + s0 - self.tau0))))

    def gradients(self, m0, s0, lam0, psi0):
        """(dKL0/dm0, dKL0/ds0) including the Lagrange multipliers at t = 0 (prior_kl0.py:94-175)."""
        z0 = m0 - self.mu0
        if self.single_dim:
            return lam0 + z0 / self.tau0, psi0 + 0.5 * (1.0 / self.tau0 - 1.0 / s0)

        def spd_inv(x):
            c_inv = np.linalg.solve(np.linalg.cholesky(x), np.eye(x.shape[0]))
            return c_inv.T.dot(c_inv)

        d_m0 = lam0 + np.linalg.solve(self.tau0, z0.T).T
        d_s0 = psi0 + 0.5 * (spd_inv(self.tau0) - spd_inv(np.asarray(s0)))
        return d_m0, d_s0
