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
