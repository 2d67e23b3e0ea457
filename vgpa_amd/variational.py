"""
VarGP: host-side mirror of src/var_bayes/variational.py (constructor, `initialization`, `free_energy`,
`gradient(x, eval_fun=False)`, `arg_out`) whose arithmetic runs fused on the GPU:

    free_energy(x) -> vgpa_free_energy : fwd sweep -> E_obs -> E_sde terms -> bwd sweep -> F
    gradient(x)    -> vgpa_gradient    : per-grid-point assembly from the state left resident in HBM
    gradient(x, eval_fun=True) -> vgpa_sweep

Like the reference the object is stateful: `gradient(x)` without `eval_fun` uses the state cached by the last
`free_energy` call (SCG relies on that ordering, src/numerics/optim_scg.py:163-167).
"""
import numpy as np
from scipy.interpolate import CubicSpline

from ._lib import Context


class VarGP(object):

    def __init__(self, model, m0, s0, fwd_ode, bwd_ode, likelihood, kl0, obs_y, obs_t, device=0, flags=0, batch=1) -> None:
        self.model = model
        self.fwd_ode, self.bwd_ode = fwd_ode, bwd_ode
        self.kl0, self.likelihood = kl0, likelihood
        self.obs_y, self.obs_t = obs_y, obs_t
        self.dt = self.model.time_step
        if self.model.single_dim:
            self.dim_n, self.dim_d = self.model.sample_path.size, 1
        else:
            self.dim_n, self.dim_d = self.model.sample_path.shape
        self.dim_tot = self.dim_n * self.dim_d * self.dim_d
        self.output = {"m0": m0, "s0": s0}
        self.device, self.flags, self.batch = device, flags, int(batch)
        self._ctx = None
        self._ctx_key = None
        self._e0 = None
        self._stale = set()
        method_f = str(getattr(fwd_ode, "method", "")).lower()
        method_b = str(getattr(bwd_ode, "method", "")).lower()
        if method_f != method_b:
            raise ValueError(f" {self.__class__.__name__}: forward ({method_f}) and backward ({method_b})"
                             f" integration methods differ; the fused sweep needs one method.")
        self._method = method_f

    # ------------------------------------------------------------------------------------------
    def _inputs(self):
        """Everything the device context bakes in at creation, read from the objects NOW."""
        single, d = self.model.single_dim, self.dim_d
        m0, s0 = self.output["m0"], self.output["s0"]
        lik = self.likelihood
        sigma = np.array([[self.model.sigma]], dtype=float) if single else np.asarray(self.model.sigma, dtype=float)
        op = getattr(lik, "operator", None)
        return dict(sigma=sigma, theta=np.atleast_1d(np.asarray(self.model.theta, dtype=float)),
                    m0=np.atleast_1d(np.asarray(m0, dtype=float)), s0=np.asarray(s0, dtype=float).reshape(d, d),
                    obs_t=np.asarray(lik.times, dtype=np.int64), obs_y=np.asarray(lik.values, dtype=float),
                    obs_noise=np.asarray(lik.noise, dtype=float).reshape(d, d),
                    # the default operator is the identity: passing None keeps libvgpa_hip on its diagonal path
                    obs_h=None if getattr(lik, "default_operator", False) else np.asarray(op, dtype=float).reshape(d, d))

    @staticmethod
    def _fingerprint(inputs):
        """Identity of the baked-in inputs: exact bytes for small arrays, a 128-bit digest of the whole buffer beyond (a D >= 257
        sigma / s0 / obs_noise is hashed in well under a millisecond -- nothing next to a sweep at that size)."""
        import hashlib
        key = []
        for name in sorted(inputs):
            a = inputs[name]
            if a is None:
                key.append((name, None))
            elif a.size <= 1 << 16:
                key.append((name, a.shape, a.tobytes()))
            else:
                key.append((name, a.shape, hashlib.blake2b(np.ascontiguousarray(a).data, digest_size=16).digest()))
        return tuple(key)

    def invalidate(self):
        """Drops the device context (and the state cached in it); the next call rebuilds it from the current
        model / likelihood / prior values."""
        if self._ctx is not None:
            self._ctx.close()
        self._ctx, self._ctx_key, self._stale = None, None, set()

    def _context(self):
        # theta, sigma, the observation noise and (m0, s0) are settable on the reference's objects
        # (stochastic_process.py / likelihood.py setters): a context built from older values must not be reused.
        inputs = self._inputs()
        key = self._fingerprint(inputs)
        if self._ctx is not None and key != self._ctx_key:
            self.invalidate()
        # E0 depends on the prior (kl0.mu0 / kl0.tau0 are plain attributes; the reference evaluates kl0 on every
        # free_energy call, variational.py:185): handed to the context per call, never baked in
        prior = tuple(np.asarray(getattr(self.kl0, k, 0.0), dtype=float).tobytes() for k in ("mu0", "tau0"))
        if self._e0 is None or self._e0[0] != (prior, key):
            self._e0 = ((prior, key), float(np.asarray(self.kl0(self.output["m0"], self.output["s0"]))))
        if self._ctx is None:
            self._ctx = Context(self.model._model_id, self._method, self.dim_d, self.dim_n, float(self.fwd_ode.dt),
                                e0=self._e0[1], device=self.device, flags=self.flags, batch=self.batch, **inputs)
            self._ctx_key = key
        self._ctx.set_prior_energy(self._e0[1])
        return self._ctx

    def initialization(self):
        """Cubic-spline initial guess of (A_t, b_t), src/var_bayes/variational.py:73-139 (host-side)."""
        tw = self.model.time_window
        knots = [tw[0], *tw[self.obs_t], tw[-1]]
        if self.model.single_dim:
            vals = np.hstack((self.obs_y[0], self.obs_y, self.obs_y[-1]))
            a0 = 0.5 * (self.model.sigma / 0.25) * np.ones(self.dim_n)
            b0 = CubicSpline(knots, vals)(tw)
        else:
            vals = np.vstack((self.obs_y[0], self.obs_y, self.obs_y[-1]))
            mt0 = CubicSpline(knots, vals)(tw)
            a0 = np.zeros((self.dim_n, self.dim_d, self.dim_d))
            b0 = np.zeros((self.dim_n, self.dim_d))
            slope = np.diff(mt0, axis=0) / self.dt
            half_k = 0.5 * np.diag(self.model.sigma.diagonal() / (0.25 * np.eye(self.dim_d)).diagonal())
            for k in range(self.dim_n - 1):
                a0[k] = half_k
                b0[k] = slope[k] + a0[k].diagonal() * mt0[k]
            a0[-1] = half_k
            b0[-1] = a0[-1].diagonal() * mt0[-1]
        return np.concatenate((a0.ravel(), b0.ravel()))

    def free_energy(self, x):
        """E0 + Esde + Eobs at x; leaves (m, S, lam, Psi, <f>) resident on the device."""
        f = self._context().free_energy(np.asarray(x, dtype=float))
        self._stale = {"mt", "st", "Efx", "Edf", "lamt", "psit"}
        return f

    def gradient(self, x, eval_fun=False):
        """Gradient of the Lagrangian w.r.t. (A_t, b_t), scaled by dt (variational.py:202-289)."""
        ctx = self._context()
        if eval_fun:
            _, g = ctx.sweep(np.asarray(x, dtype=float))
            self._stale = {"mt", "st", "Efx", "Edf", "lamt", "psit"}
            return g
        return ctx.gradient(None)

    def sweep(self, x):
        """(F, grad) in one call: what SCG's df(x, eval_fun=True) evaluates."""
        f, g = self._context().sweep(np.asarray(x, dtype=float))
        self._stale = {"mt", "st", "Efx", "Edf", "lamt", "psit"}
        return f, g

    def device_scg(self, *options):
        """SCG with x, d and the gradients resident in HBM (scg.DeviceSCG); `batch` problems advance in lock step."""
        from .scg import DeviceSCG
        self._stale = {"mt", "st", "Efx", "Edf", "lamt", "psit"}
        return DeviceSCG(self._context(), *options)

    @property
    def arg_out(self):
        """The reference's output dictionary (m0, s0, mt, st, Efx, Edf, lamt, psit), fetched lazily."""
        for key in sorted(self._stale):
            val = self._ctx.fetch(key)
            if self.model.single_dim:
                val = val.reshape(self.dim_n)
            self.output[key] = val
        self._stale = set()
        return self.output
