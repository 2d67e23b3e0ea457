"""
Scaled conjugate gradients: the caller of the hot path (the reference drives `VarGP.free_energy / gradient` with
NETLAB's `scg`, src/numerics/optim_scg.py:75-285).

One engine, `_lock_step`, advances B independent minimisations together: every per-problem scalar of the method
(mu, kappa, theta, delta, alpha, beta, gamma, f) is a (B,) array, every decision of the method is a boolean mask, and
the vectors (x, d, two gradients, two trial vectors) live wherever the *backend* keeps them:

  * `_DeviceVectors` -- segments of DeviceBuffers in HBM, algebra by vecops.hip, objective by the `*_dev` entry points
    (x never crosses PCIe; only the (B,) scalars do).  `DeviceSCG` is this backend: the shipped optimiser.
  * `_HostVectors` -- numpy rows and a user-supplied pair `f(x)`, `df(x, eval_fun=False)`.  `SCG(f, df, options)` is
    this backend with B = 1; it exists so that code written against the reference's optimiser interface (and the
    reference's recorded optimisation traces, tests/golden/scg_*.json) runs unchanged against the GPU objective.

A problem that has met its stopping rule gets zero coefficients from then on (its vectors are frozen) while the rest of
the batch continues.  The order in which the objective is evaluated -- f and g at x0, g at x + sigma d, f at x + alpha d,
g at an accepted point -- is the method's; the host backend additionally repeats `f(x)` at an accepted point because the
reference does (optim_scg.py:232-235) and its evaluation counts are part of the recorded traces, whereas the device
backend assembles that gradient from the state `f(x + alpha d)` left in HBM (one sweep less, same numbers).
"""
import numpy as np

_SIGMA0 = 1.0e-3                       # finite-difference scale of the curvature probe
_BETA_MIN, _BETA_MAX = 1.0e-15, 1.0e+100


def _options(args):
    opts = dict(args[0]) if args else {}
    return (int(opts.get("max_it", 150)), float(opts.get("x_tol", 1.0e-6)), float(opts.get("f_tol", 1.0e-8)),
            bool(opts.get("display", False)))


def _new_stats(nit, nb):
    return {"MaxIt": np.full(nb, nit), "fx": np.zeros((nit, nb)), "dfx": np.zeros((nit, nb)),
            "f_eval": 0.0, "df_eval": 0.0, "beta": np.zeros((nit, nb))}


class _HostVectors:
    """numpy rows + callables.  Vectors are (B, n) arrays; B = 1 unless `f`/`df` are themselves batched."""

    def __init__(self, f, df, extra=()):
        self.f, self.df, self.extra, self.B = f, df, tuple(extra), 1

    def vectors(self, x0, count):
        x0 = np.asarray(x0, dtype=float).reshape(self.B, -1)
        return [x0.copy()] + [np.zeros_like(x0) for _ in range(count - 1)]

    def release(self, vecs):
        pass

    def _each(self, fn, x, *a, **k):
        return np.stack([np.asarray(fn(row, *a, **k), dtype=float) for row in x])

    def value_and_gradient(self, x, g, st):
        f = self._each(self.f, x, *self.extra).reshape(self.B)
        g[...] = self._each(self.df, x, *self.extra)
        st["f_eval"] += 1
        st["df_eval"] += 1
        return f

    def probe_gradient(self, x, g, st):
        g[...] = self._each(self.df, x, eval_fun=True)
        st["f_eval"] += 1
        st["df_eval"] += 1

    def value(self, x, st):
        st["f_eval"] += 1
        return self._each(self.f, x, *self.extra).reshape(self.B)

    def gradient_at_accepted(self, x, g, st):
        self.value_and_gradient(x, g, st)

    def dot(self, a, b):
        return np.array([ra.dot(rb) for ra, rb in zip(a, b)])

    def absmax(self, a):
        return np.abs(a).max(axis=1)

    def asum(self, a):
        return np.abs(a).sum(axis=1)

    def axpby(self, al, a, be, b, out):
        al = np.broadcast_to(np.asarray(al, dtype=float), (self.B,))[:, None]
        if b is None:
            out[...] = al * a
        else:
            be = np.broadcast_to(np.asarray(be, dtype=float), (self.B,))[:, None]
            # a zero coefficient must not touch its operand (an evaluated trial point may hold inf / nan)
            out[...] = np.where(al != 0.0, al * a, 0.0) + np.where(be != 0.0, be * b, 0.0)

    def fetch(self, x):
        return np.array(x)


class _DeviceVectors:
    """B segments per DeviceBuffer on the context's GPU; see vgpa_amd/csrc/vecops.hip."""

    def __init__(self, ctx):
        self.ctx, self.B = ctx, ctx.B

    def vectors(self, x0, count):
        from ._lib import DeviceBuffer
        x0 = np.ascontiguousarray(np.asarray(x0, dtype=float)).reshape(self.B, -1)
        vecs = [DeviceBuffer(self.ctx, x0.size) for _ in range(count)]
        vecs[0].upload(x0)
        return vecs

    def release(self, vecs):
        self.ctx.synchronize()
        self.ctx.release_x()          # the context must not keep pointing into a trial vector that is about to go
        for v in vecs:
            v.free()

    def value_and_gradient(self, x, g, st):
        st["f_eval"] += 1
        st["df_eval"] += 1
        return np.atleast_1d(self.ctx.sweep_dev(x, g)).astype(float)

    def probe_gradient(self, x, g, st):
        self.value_and_gradient(x, g, st)

    def value(self, x, st):
        st["f_eval"] += 1
        return np.atleast_1d(self.ctx.free_energy_dev(x)).astype(float)

    def gradient_at_accepted(self, x, g, st):
        st["df_eval"] += 1
        self.ctx.gradient_dev(g)       # from the state value(x) left resident

    def dot(self, a, b):
        return self.ctx.vdot(a, b)

    def absmax(self, a):
        return self.ctx.vabsmax(a)

    def asum(self, a):
        return self.ctx.vasum(a)

    def axpby(self, al, a, be, b, out):
        self.ctx.vaxpby(al, a, be, b, out)

    def fetch(self, x):
        return x.download().reshape(self.B, -1)


def _lock_step(vs, x0, nit, x_tol, f_tol, display, st):
    """Runs the method on backend `vs`; returns (x (B, n), f (B,))."""
    vecs = vs.vectors(x0, 6)
    try:
        return _iterate(vs, vecs, nit, x_tol, f_tol, display, st)
    finally:
        vs.release(vecs)


def _iterate(vs, vecs, nit, x_tol, f_tol, display, st):
    x, x_try, d, g_new, g_old, g_tmp = vecs
    nb = vs.B
    eps = np.finfo(float).eps
    ones, zeros = np.ones(nb), np.zeros(nb)
    pick = lambda mask: np.where(mask, 1.0, 0.0)         # noqa: E731  coefficient 1 where the mask holds
    keep = lambda mask: np.where(mask, 0.0, 1.0)         # noqa: E731  ... and its complement

    f_now = vs.value_and_gradient(x, g_new, st)
    f_old = f_now.copy()
    vs.axpby(ones, g_new, None, None, g_old)
    vs.axpby(-ones, g_new, None, None, d)
    n_par = _segment_length(vs, x)
    ok = np.ones(nb, dtype=bool)                 # last trial step was accepted
    done = np.zeros(nb, dtype=bool)
    n_ok = np.zeros(nb, dtype=int)               # accepted steps since the last restart of the direction
    beta = np.ones(nb)                           # trust-region scale
    kappa, theta, mu = np.zeros(nb), np.zeros(nb), np.zeros(nb)
    f_ret = f_now.copy()

    def stop(mask, value, j):
        nonlocal done
        mask = mask & ~done
        f_ret[mask] = value[mask]
        st["MaxIt"][mask] = j + 1
        done = done | mask

    for j in range(nit):
        act = ok & ~done
        if act.any():                            # second-order information along d (only after an accepted step)
            mu_c = vs.dot(d, g_new)
            uphill = act & (mu_c >= 0.0)
            if uphill.any():
                vs.axpby(-pick(uphill), g_new, keep(uphill), d, d)
                mu_c = vs.dot(d, g_new)
            mu = np.where(act, mu_c, mu)
            kappa = np.where(act, vs.dot(d, d), kappa)
            stop(act & (kappa < eps), f_now, j)
            act = ok & ~done
            if done.all():
                break
            with np.errstate(divide="ignore", invalid="ignore"):
                sigma = np.where(act, _SIGMA0 / np.sqrt(kappa), 0.0)
            vs.axpby(ones, x, sigma, d, x_try)
            vs.probe_gradient(x_try, g_tmp, st)
            vs.axpby(ones, g_tmp, -ones, g_new, g_tmp)
            with np.errstate(divide="ignore", invalid="ignore"):
                theta = np.where(act, vs.dot(d, g_tmp) / sigma, theta)

        live = ~done
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = theta + beta * kappa
            indefinite = live & (delta <= 0.0)
            delta = np.where(indefinite, beta * kappa, delta)
            beta = np.where(indefinite, beta - theta / kappa, beta)
            alpha = np.where(live, -(mu / delta), 0.0)

        vs.axpby(ones, x, alpha, d, x_try)
        f_new = vs.value(x_try, st)
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = 2.0 * (f_new - f_old) / (alpha * mu)          # actual vs predicted decrease
        succ = live & (delta >= 0.0)
        fail = live & ~succ
        # the recorded gradient norm is the one in force *before* the refresh below (g_now of optim_scg.py:208,213)
        total_grad = np.where(succ, vs.asum(g_new), vs.asum(g_old))
        n_ok = n_ok + succ
        ok = np.where(live, succ, ok)
        f_now = np.where(succ, f_new, np.where(fail, f_old, f_now))
        st["fx"][j], st["beta"][j], st["dfx"][j] = f_now, beta, total_grad
        if display and j % 10 == 0:
            print(" {0}: fx={1:.3f}\tsum(gx)={2:.3f}".format(j, f_now.sum(), total_grad.sum()))

        if succ.any():
            vs.axpby(pick(succ), x_try, keep(succ), x, x)          # x <- x_try where accepted
            step = np.abs(alpha) * vs.absmax(d)
            stop(succ & (step <= x_tol) & (np.abs(f_new - f_old) <= f_tol), f_new, j)
            go = succ & ~done
            if go.any():
                f_old = np.where(go, f_new, f_old)
                vs.axpby(pick(go), g_new, keep(go), g_old, g_old)
                vs.gradient_at_accepted(x, g_tmp, st)
                vs.axpby(pick(go), g_tmp, keep(go), g_new, g_new)
                stop(go & np.isclose(vs.dot(g_new, g_new), 0.0), f_now, j)
        if done.all():
            break

        live = ~done
        beta = np.where(live & (delta < 0.25), np.minimum(4.0 * beta, _BETA_MAX), beta)
        beta = np.where(live & (delta > 0.75), np.maximum(0.5 * beta, _BETA_MIN), beta)

        restart = live & (n_ok == n_par)
        conj = live & ~restart & ok
        if restart.any() or conj.any():
            gamma = zeros
            if conj.any():
                vs.axpby(ones, g_old, -ones, g_new, g_tmp)
                with np.errstate(divide="ignore", invalid="ignore"):
                    gamma = np.maximum(vs.dot(g_new, g_tmp) / mu, 0.0)
            vs.axpby(np.where(conj, gamma, keep(restart)), d, -pick(conj | restart), g_new, d)
            n_ok = np.where(restart, 0, n_ok)

    if not done.all():
        print(" SCG: iteration limit reached before the stopping rule was met.")
        f_ret[~done] = f_old[~done]
    return vs.fetch(x), f_ret


def _segment_length(vs, x):
    if isinstance(x, np.ndarray):
        return x.shape[1]
    return x.count // vs.B


class DeviceSCG(object):
    """SCG over a batched `Context` with every vector resident in HBM (`VarGP.device_scg`).  `statistics` holds
    (max_it, B) traces; `f_eval` / `df_eval` count device evaluations of the whole batch."""

    def __init__(self, ctx, *args) -> None:
        self.ctx = ctx
        self.nit, self.x_tol, self.f_tol, self.display = _options(args)
        self.stats = _new_stats(self.nit, ctx.B)

    @property
    def statistics(self) -> dict:
        return self.stats

    def __call__(self, x0):
        x, f = _lock_step(_DeviceVectors(self.ctx), x0, self.nit, self.x_tol, self.f_tol, self.display, self.stats)
        if self.ctx.B == 1:
            return x[0], float(f[0])
        return x, f


class SCG(object):
    """`SCG(f, df, options)(x0, *args) -> (x, fx)`: the optimiser interface the reference's driver uses
    (optim_scg.py:23-73), served by the lock-step engine with host vectors and one problem."""

    def __init__(self, f, df, *args) -> None:
        self.f, self.df = f, df
        self.nit, self.x_tol, self.f_tol, self.display = _options(args)
        self.stats = _new_stats(self.nit, 1)
        self._squeeze()

    def _squeeze(self):
        st = self.stats
        for key in ("fx", "dfx", "beta"):
            st[key] = np.asarray(st[key]).reshape(self.nit)
        st["MaxIt"] = int(np.asarray(st["MaxIt"]).ravel()[0])

    @property
    def statistics(self) -> dict:
        return self.stats

    def __call__(self, x0, *args):
        self.stats = _new_stats(self.nit, 1)
        try:
            x, f = _lock_step(_HostVectors(self.f, self.df, args), np.asarray(x0, dtype=float).ravel(), self.nit,
                              self.x_tol, self.f_tol, self.display, self.stats)
        finally:
            self._squeeze()
        return x[0], float(f[0])
