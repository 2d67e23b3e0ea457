"""
Scaled Conjugate Gradient driver: the caller of the hot path (src/numerics/optim_scg.py:23-285, NETLAB's
`scg`).  Host Python by design (BASELINE.json north_star); it calls `f(x)`, `df(x)` and
`df(x_plus, eval_fun=True)` in the same order as the reference so that an optimisation trace is comparable
evaluation by evaluation.
"""
import numpy as np


class SCG(object):

    def __init__(self, f, df, *args) -> None:
        opts = args[0] if args else {}
        self.f, self.df = f, df
        self.nit = opts.get("max_it", 150)
        self.x_tol = opts.get("x_tol", 1.0e-6)
        self.f_tol = opts.get("f_tol", 1.0e-8)
        self.display = opts.get("display", False)
        self.stats = {"MaxIt": self.nit, "fx": np.zeros(self.nit), "dfx": np.zeros(self.nit),
                      "f_eval": 0.0, "df_eval": 0.0, "beta": np.zeros(self.nit)}

    @property
    def statistics(self) -> dict:
        return self.stats

    def __call__(self, x0: np.ndarray, *args):
        st = self.stats
        x = x0.flatten()
        n_par = x.size
        sigma0 = 1.0e-3
        beta, beta_lo, beta_hi = 1.0, 1.0e-15, 1.0e+100
        eps = np.finfo(float).eps

        f_now = self.f(x, *args)
        g_new = self.df(x, *args)
        st["f_eval"] += 1
        st["df_eval"] += 1
        f_old, g_old = f_now, np.copy(g_new)
        d = -g_new
        ok, n_ok = True, 0
        kappa = theta = mu = 0.0

        for j in range(self.nit):
            if ok:
                # directional derivative and curvature along d
                mu = d.T.dot(g_new)
                if mu >= 0.0:
                    d = -g_new
                    mu = d.T.dot(g_new)
                kappa = d.T.dot(d)
                if kappa < eps:
                    st["MaxIt"] = j + 1
                    return x, f_now
                sigma = sigma0 / np.sqrt(kappa)
                g_plus = self.df(x + (sigma * d), eval_fun=True)
                st["f_eval"] += 1
                st["df_eval"] += 1
                theta = (d.T.dot(g_plus - g_new)) / sigma

            delta = theta + (beta * kappa)
            if delta <= 0.0:
                delta = beta * kappa
                beta = beta - (theta / kappa)
            alpha = -(mu / delta)

            x_new = x + (alpha * d)
            f_new = self.f(x_new, *args)
            st["f_eval"] += 1
            delta = 2.0 * (f_new - f_old) / (alpha * mu)     # comparison ratio
            if delta >= 0.0:
                ok = True
                n_ok += 1
                x, f_now, g_now = np.copy(x_new), np.copy(f_new), np.copy(g_new)
            else:
                ok = False
                f_now, g_now = f_old, np.copy(g_old)

            total_grad = np.sum(np.abs(g_now))
            st["fx"][j], st["beta"][j], st["dfx"][j] = f_now, beta, total_grad
            if self.display and (np.mod(j, 10) == 0):
                print(" {0}: fx={1:.3f}\tsum(gx)={2:.3f}".format(j, f_now, total_grad))

            if ok:
                if (np.abs(alpha * d).max() <= self.x_tol) and (np.abs(f_new - f_old) <= self.f_tol):
                    st["MaxIt"] = j + 1
                    return x, f_new
                f_old, g_old = f_new, np.copy(g_new)
                f_now = self.f(x, *args)
                g_new = self.df(x, *args)
                st["f_eval"] += 1
                st["df_eval"] += 1
                if np.isclose(g_new.T.dot(g_new), 0.0):
                    st["MaxIt"] = j + 1
                    return x, f_now

            if delta < 0.25:
                beta = np.minimum(4.0 * beta, beta_hi)
            if delta > 0.75:
                beta = np.maximum(0.5 * beta, beta_lo)

            if n_ok == n_par:
                d = -g_new
                n_ok = 0
            elif ok:
                gamma = np.maximum(g_new.T.dot(g_old - g_new) / mu, 0.0)
                d = (gamma * d) - g_new

        print(" SGC: Maximum number of iterations has been reached.")
        return x, f_old


class DeviceSCG(object):
    """
    The same SCG iteration (src/numerics/optim_scg.py:75-285) with every vector resident in HBM
    and advanced in lock step for the `B` independent problems of a batched `Context`:

      * x, d, g_new, g_old and the two trial vectors are DeviceBuffers; the host sees only the per-problem scalars
        (mu, kappa, theta, delta, alpha, gamma, f) that drive the unchanged control flow, here vectorised over the
        batch with masks.  A problem that has converged gets zero coefficients and is frozen.
      * the objective / gradient calls are the context's `*_dev` entry points, so x never crosses PCIe;
      * the reference re-evaluates `f(x)` right after accepting `x = x_new` (optim_scg.py:232-235) although the state
        of `f(x_new)` is still cached; here the gradient is assembled from that cached state (one forward-backward
        sweep less per successful iteration, identical numbers).

    `stats["f_eval"]`/`["df_eval"]` count device evaluations of the whole batch.
    """

    def __init__(self, ctx, *args) -> None:
        opts = args[0] if args else {}
        self.ctx = ctx
        self.nit = opts.get("max_it", 150)
        self.x_tol = opts.get("x_tol", 1.0e-6)
        self.f_tol = opts.get("f_tol", 1.0e-8)
        self.display = opts.get("display", False)
        b = ctx.B
        self.stats = {"MaxIt": np.full(b, self.nit), "fx": np.zeros((self.nit, b)), "dfx": np.zeros((self.nit, b)),
                      "f_eval": 0.0, "df_eval": 0.0, "beta": np.zeros((self.nit, b))}

    @property
    def statistics(self) -> dict:
        return self.stats

    def __call__(self, x0):
        from ._lib import DeviceBuffer
        ctx, st = self.ctx, self.stats
        nb = ctx.B
        x0 = np.ascontiguousarray(np.asarray(x0, dtype=float)).reshape(nb, -1)
        n_par = x0.shape[1]
        bufs = [DeviceBuffer(ctx, nb * n_par) for _ in range(6)]
        try:
            return self._run(x0, n_par, *bufs)
        finally:
            ctx.synchronize()
            for buf in bufs:
                buf.free()

    def _run(self, x0, n_par, x, x_try, d, g_new, g_old, g_tmp):
        ctx, st = self.ctx, self.stats
        nb = ctx.B
        sigma0 = 1.0e-3
        beta = np.ones(nb)
        beta_lo, beta_hi = 1.0e-15, 1.0e+100
        eps = np.finfo(float).eps
        ones, zeros = np.ones(nb), np.zeros(nb)

        x.upload(x0)
        f_now = np.atleast_1d(ctx.sweep_dev(x, g_new)).astype(float)
        st["f_eval"] += 1
        st["df_eval"] += 1
        f_old = f_now.copy()
        ctx.vaxpby(ones, g_new, None, None, g_old)
        ctx.vaxpby(-ones, g_new, None, None, d)
        ok = np.ones(nb, dtype=bool)
        done = np.zeros(nb, dtype=bool)
        n_ok = np.zeros(nb, dtype=int)
        kappa, theta, mu = np.zeros(nb), np.zeros(nb), np.zeros(nb)
        f_ret = f_now.copy()

        def finish(mask, value, j):
            nonlocal done
            mask = mask & ~done
            f_ret[mask] = value[mask]
            st["MaxIt"][mask] = j + 1
            done = done | mask

        for j in range(self.nit):
            act = ok & ~done
            if act.any():
                mu_c = ctx.vdot(d, g_new)
                reset = act & (mu_c >= 0.0)
                if reset.any():
                    ctx.vaxpby(np.where(reset, -1.0, 0.0), g_new, np.where(reset, 0.0, 1.0), d, d)
                    mu_c = ctx.vdot(d, g_new)
                mu = np.where(act, mu_c, mu)
                kappa = np.where(act, ctx.vdot(d, d), kappa)
                finish(act & (kappa < eps), f_now, j)
                act = ok & ~done
                if done.all():
                    break
                with np.errstate(divide="ignore", invalid="ignore"):
                    sigma = np.where(act, sigma0 / np.sqrt(kappa), 0.0)
                ctx.vaxpby(ones, x, sigma, d, x_try)
                ctx.sweep_dev(x_try, g_tmp)
                st["f_eval"] += 1
                st["df_eval"] += 1
                ctx.vaxpby(ones, g_tmp, -ones, g_new, g_tmp)
                with np.errstate(divide="ignore", invalid="ignore"):
                    theta = np.where(act, ctx.vdot(d, g_tmp) / sigma, theta)

            live = ~done
            with np.errstate(divide="ignore", invalid="ignore"):
                delta = theta + beta * kappa
                fix = live & (delta <= 0.0)
                delta = np.where(fix, beta * kappa, delta)
                beta = np.where(fix, beta - theta / kappa, beta)
                alpha = np.where(live, -(mu / delta), 0.0)

            ctx.vaxpby(ones, x, alpha, d, x_try)
            f_new = np.atleast_1d(ctx.free_energy_dev(x_try)).astype(float)
            st["f_eval"] += 1
            with np.errstate(divide="ignore", invalid="ignore"):
                delta = 2.0 * (f_new - f_old) / (alpha * mu)
            succ = live & (delta >= 0.0)
            fail = live & ~succ
            # statistics use the gradient *before* it is refreshed (g_now of optim_scg.py:208,213)
            total_grad = np.where(succ, ctx.vasum(g_new), ctx.vasum(g_old))
            n_ok = n_ok + succ
            ok = np.where(live, succ, ok)
            f_now = np.where(succ, f_new, np.where(fail, f_old, f_now))
            st["fx"][j], st["beta"][j], st["dfx"][j] = f_now, beta, total_grad
            if self.display and (np.mod(j, 10) == 0):
                print(" {0}: fx={1:.3f}\tsum(gx)={2:.3f}".format(j, f_now.sum(), total_grad.sum()))

            if succ.any():
                # x <- x_new where accepted.  The state cached by free_energy(x_try) is the state at the new x.
                ctx.vaxpby(np.where(succ, 1.0, 0.0), x_try, np.where(succ, 0.0, 1.0), x, x)
                step = np.abs(alpha) * ctx.vabsmax(d)
                finish(succ & (step <= self.x_tol) & (np.abs(f_new - f_old) <= self.f_tol), f_new, j)
                go = succ & ~done
                if go.any():
                    f_old = np.where(go, f_new, f_old)
                    ctx.vaxpby(np.where(go, 1.0, 0.0), g_new, np.where(go, 0.0, 1.0), g_old, g_old)
                    ctx.gradient_dev(g_tmp)
                    st["df_eval"] += 1
                    ctx.vaxpby(np.where(go, 1.0, 0.0), g_tmp, np.where(go, 0.0, 1.0), g_new, g_new)
                    gg = ctx.vdot(g_new, g_new)
                    finish(go & np.isclose(gg, 0.0), f_now, j)
            if done.all():
                break

            live = ~done
            beta = np.where(live & (delta < 0.25), np.minimum(4.0 * beta, beta_hi), beta)
            beta = np.where(live & (delta > 0.75), np.maximum(0.5 * beta, beta_lo), beta)

            restart = live & (n_ok == n_par)
            conj = live & ~restart & ok
            if restart.any() or conj.any():
                gamma = zeros
                if conj.any():
                    ctx.vaxpby(ones, g_old, -ones, g_new, g_tmp)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        gamma = np.maximum(ctx.vdot(g_new, g_tmp) / mu, 0.0)
                a_d = np.where(conj, gamma, np.where(restart, 0.0, 1.0))
                a_g = np.where(conj | restart, -1.0, 0.0)
                ctx.vaxpby(a_d, d, a_g, g_new, d)
                n_ok = np.where(restart, 0, n_ok)

        if not done.all():
            print(" SGC: Maximum number of iterations has been reached.")
            f_ret[~done] = f_old[~done]
        x_out = x.download().reshape(nb, n_par)
        if nb == 1:
            return x_out[0], float(f_ret[0])
        return x_out, f_ret
