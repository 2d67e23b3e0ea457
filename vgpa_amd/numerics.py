"""
Host-side mirror of the reference's stepper plugin surface, backed by the HIP kernels.

  num_integration = {"euler", "heun", "rk2", "rk4"}          src/numerics/utilities.py:12-13
  <Stepper>(dt, single_dim).solve_fwd / .solve_bwd           src/numerics/{euler,heun,runge_kutta2,runge_kutta4}.py
  FwdOde(dt, method, single_dim)(at, bt, m0, s0, sigma)      src/var_bayes/fwd_ode.py:14-65
  BwdOde(dt, method, single_dim)(at, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds)   src/var_bayes/bwd_ode.py:14-65

Same names, argument meaning and error behaviour (ValueError for dt <= 0 and for an unknown method).
Every call goes through libvgpa_hip.so (vgpa_solve_fwd / vgpa_solve_bwd); there is no CPU path.
"""
import numpy as np

from ._lib import Context

SMALL_D_MAX = 64      # D <= 64: one-workgroup LDS-resident kernels; above: per-stage GEMM path


class OdeSolver(object):
    """Parent of the four steppers (src/numerics/ode_solver.py:4-29)."""

    method = None

    def __init__(self, dt: float = 0.01, single_dim: bool = True) -> None:
        if dt <= 0.0:
            raise ValueError(f" {self.__class__.__name__}:"
                             f" Discrete time step should be positive --> {dt}.")
        self.dt = dt
        self.single_dim = single_dim
        self._ctx = {}
        self.device = 0
        self.flags = 0

    @staticmethod
    def _sharded():
        """True when a torch.distributed process group with more than one rank is active (row-sharded large-D path)."""
        import sys
        if "torch" not in sys.modules:
            return False
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _drop_contexts(self):
        """One live workspace per stepper object.  A row-sharded recursion owns an RCCL communicator: it is closed HERE, in program
        order (the same on every rank), not whenever the garbage collector gets to it."""
        for obj in self._ctx.values():
            close = getattr(obj, "close", None)
            if close is not None:
                close()
        self._ctx.clear()

    def _large(self, dim_d, n_pts):
        """D > 64 under a live process group: the row-sharded native driver (vgpa_amd/large_d.py) on the default group.  The driver
        shards rows evenly: D must be a multiple of the number of ranks -- otherwise None, and every rank runs the recursion by
        itself on its own GPU (same results, no speed-up)."""
        import torch.distributed as dist
        if dim_d % dist.get_world_size():
            return None
        from .large_d import NativeShardedRecursion
        key = ("large", dim_d, n_pts)
        rec = self._ctx.get(key)
        if rec is None:
            self._drop_contexts()
            rec = NativeShardedRecursion(self.method, self.dt, dim_d, n_pts)
            self._ctx[key] = rec
        return rec

    @staticmethod
    def _gather_time(rec, v_own, m_own, n_pts):
        """The driver's results are time-sharded; the reference's contract is the whole grid on the caller: one all-gather of the
        vector slices and one of the matrix slices into preallocated tensors (slices padded to the longest; nothing is pickled)."""
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        d, world = rec.D, rec.world
        lo, hi = rec.time_slice
        mine = torch.tensor([lo, hi], dtype=torch.int64, device=dev)
        slices = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(slices, mine)
        slices = [tuple(int(x) for x in s.cpu()) for s in slices]
        n_max = max(h - l for l, h in slices)
        pv, pm = torch.zeros((n_max, d), dtype=torch.float64, device=dev), torch.zeros((n_max, d, d), dtype=torch.float64, device=dev)
        pv[:hi - lo] = torch.from_numpy(np.ascontiguousarray(v_own.numpy()).reshape(hi - lo, d)).to(dev)
        pm[:hi - lo] = torch.from_numpy(np.ascontiguousarray(m_own.numpy()).reshape(hi - lo, d, d)).to(dev)
        gv, gm = [torch.empty_like(pv) for _ in range(world)], [torch.empty_like(pm) for _ in range(world)]
        dist.all_gather(gv, pv)
        dist.all_gather(gm, pm)
        v, m = np.empty((n_pts, d)), np.empty((n_pts, d, d))
        for (l, h), tv, tm in zip(slices, gv, gm):
            v[l:h], m[l:h] = tv[:h - l].cpu().numpy(), tm[:h - l].cpu().numpy()
        return v, m

    def _context(self, dim_d, n_pts):
        key = (dim_d, n_pts, self.device, self.flags)
        ctx = self._ctx.get(key)
        if ctx is None:
            self._drop_contexts()      # one live workspace per stepper object
            ctx = Context("NONE", self.method, dim_d, n_pts, self.dt, sigma=np.eye(dim_d), device=self.device,
                          flags=self.flags)
            self._ctx[key] = ctx
        return ctx

    def solve_fwd(self, lin_a, off_b, m0, s0, sigma):
        """(m_t, S_t): same contract as <stepper>.solve_fwd of the reference."""
        off_b = np.asarray(off_b, dtype=float)
        if self.single_dim:
            n = off_b.size
            ctx = self._context(1, n)
            mt, st = ctx.solve_fwd(np.asarray(lin_a, dtype=float).reshape(n, 1, 1), off_b.reshape(n, 1),
                                   np.array([m0], dtype=float), np.array([s0], dtype=float),
                                   np.array([sigma], dtype=float))
            return mt.reshape(n), st.reshape(n)
        n, d = off_b.shape
        rec = self._large(d, n) if (d > SMALL_D_MAX and self._sharded()) else None
        if rec is not None:
            return self._gather_time(rec, *rec.solve_fwd(lin_a, off_b, m0, s0, sigma), n)
        ctx = self._context(d, n)
        return ctx.solve_fwd(lin_a, off_b, m0, s0, sigma)

    def solve_bwd(self, lin_a, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds):
        """(lam_t, Psi_t): same contract as <stepper>.solve_bwd of the reference."""
        dEsde_dm = np.asarray(dEsde_dm, dtype=float)
        if self.single_dim:
            n = dEsde_dm.size
            ctx = self._context(1, n)
            r3 = lambda v: np.asarray(v, dtype=float).reshape(n, 1, 1)   # noqa: E731
            r2 = lambda v: np.asarray(v, dtype=float).reshape(n, 1)      # noqa: E731
            lam, psi = ctx.solve_bwd(r3(lin_a), r2(dEsde_dm), r3(dEsde_ds), r2(dEobs_dm), r3(dEobs_ds))
            return lam.reshape(n), psi.reshape(n)
        n, d = dEsde_dm.shape
        rec = self._large(d, n) if (d > SMALL_D_MAX and self._sharded()) else None
        if rec is not None:
            return self._gather_time(rec, *rec.solve_bwd(lin_a, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds), n)
        ctx = self._context(d, n)
        return ctx.solve_bwd(lin_a, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds)


class Euler(OdeSolver):
    method = "euler"


class Heun(OdeSolver):
    method = "heun"


class RungeKutta2(OdeSolver):
    method = "rk2"


class RungeKutta4(OdeSolver):
    method = "rk4"


# Registry with the reference's keys (src/numerics/utilities.py:12-13).
num_integration = {"euler": Euler, "heun": Heun, "rk2": RungeKutta2, "rk4": RungeKutta4}


class _OdeFunctor(object):
    def __init__(self, dt: float, method: str, single_dim: bool = True) -> None:
        if dt <= 0.0:
            raise ValueError(f" {self.__class__.__name__}:"
                             f" Discrete time step should be strictly positive -> {dt}.")
        self.dt = dt
        method_str = str(method).lower()
        try:
            self.solver = num_integration[method_str](dt, single_dim)
        except KeyError:
            raise ValueError(f" {self.__class__.__name__}:"
                             f" Integration method is unknown -> {method}.")
        self.method = method

    def __str__(self) -> str:
        return f" {self.__class__.__name__} Id({id(self)}): dt={self.dt}, method={self.method}"


class FwdOde(_OdeFunctor):
    """Forward ODE functor (src/var_bayes/fwd_ode.py)."""

    def __call__(self, at, bt, m0, s0, sigma):
        return self.solver.solve_fwd(at, bt, m0, s0, sigma)


class BwdOde(_OdeFunctor):
    """Backward ODE functor (src/var_bayes/bwd_ode.py)."""

    def __call__(self, at, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds):
        return self.solver.solve_bwd(at, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds)
