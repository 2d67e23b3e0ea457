"""
TEST INFRASTRUCTURE (not shipped): the round-1 Python statement of the row-sharded schedule -- the two stage kernels issued
stage by stage from Python with torch.distributed collectives.  The product's driver is native (vgpa_shard_*,
vgpa_amd/large_d.py::NativeShardedRecursion); this one stays here because a CPU stand-in for the two kernels lets the
sharding / collective logic run under gloo with world size 2 in the CPU suite, and because `HipStageBackend` is a convenient
handle on vgpa_ld_gemm / vgpa_ld_stage for the GEMM tests.
"""
import ctypes

import numpy as np

from vgpa_amd._lib import load, LdStageArgs, _raise
from vgpa_amd.parallel import shard_range

METHODS = ("euler", "heun", "rk2", "rk4")


class HipStageBackend:
    """The product backend: launches the HIP kernels on torch's current stream."""

    def __init__(self):
        self._lib = load()

    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _p(t, offset=0):
        return None if t is None else ctypes.c_void_p(t.data_ptr() + 8 * int(offset))

    def gemm(self, transa, M, N, K, A0, a0_off, A1, a1_off, lda, B, ldb, C, cw):
        rc = self._lib.vgpa_ld_gemm(self._stream(), int(transa), M, N, K, self._p(A0, a0_off), self._p(A1, a1_off), lda,
                                    self._p(B), ldb, self._p(C), cw)
        if rc != 0:
            _raise(rc, "vgpa_ld_gemm failed")

    def stage(self, **kw):
        a = LdStageArgs()
        for name in ("D", "row0", "Mp", "cw", "fwd", "kstore", "final_mode", "lda"):
            setattr(a, name, int(kw[name]))
        a.cx, a.cf = float(kw["cx"]), float(kw["cf"])
        for name in ("W", "Wcol", "E0", "E1", "J", "base", "K1", "K23", "out", "A0", "A1", "x", "e0", "e1", "jv",
                     "vbase", "k1v", "k23v", "vout"):
            v = kw.get(name)
            if isinstance(v, tuple):
                setattr(a, name, self._p(v[0], v[1]))
            else:
                setattr(a, name, self._p(v))
        rc = self._lib.vgpa_ld_stage(self._stream(), ctypes.byref(a))
        if rc != 0:
            _raise(rc, "vgpa_ld_stage failed")


def _is_symmetric(t):
    import torch
    scale = float(t.abs().max())
    return scale == 0.0 or float((t - t.transpose(-1, -2)).abs().max()) <= 1e-14 * scale


class ShardedRecursion:
    """(m_t, S_t) and (lam_t, Psi_t) for D > 64 on one GPU or row-sharded over a process group."""

    def __init__(self, method, dt, dim_d, group=None, backend=None, device=None, comm=None):
        """`comm`: object with torch.distributed's get_world_size / get_rank / all_to_all_single /
        all_gather_into_tensor (default: torch.distributed itself when a process group is initialised)."""
        import torch
        import torch.distributed as dist
        method = str(method).lower()
        if method not in METHODS:
            raise ValueError(f" Integration method is unknown -> {method}.")
        if dt <= 0.0:
            raise ValueError(f" Discrete time step should be strictly positive -> {dt}.")
        self.method, self.dt, self.D = method, float(dt), int(dim_d)
        self.group = group
        self.dist = comm if comm is not None else (dist if (dist.is_available() and dist.is_initialized()) else None)
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        if self.D % self.world != 0:
            raise ValueError(f"D={self.D} must be a multiple of the number of ranks ({self.world})")
        self.row0, hi = shard_range(self.D, self.rank, self.world)
        self.Mp = hi - self.row0
        self.cw = self.Mp
        self.backend = backend if backend is not None else HipStageBackend()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if isinstance(self.backend, HipStageBackend) \
                else torch.device("cpu")
        self.device = device
        f64 = dict(dtype=torch.float64, device=device)
        D, Mp = self.D, self.Mp
        self.Wp = torch.zeros(self.world * Mp * self.cw, **f64)       # [q][Mp][cw]
        self.Wcol = torch.zeros(D * Mp, **f64) if self.world > 1 else self.Wp
        self.K1, self.K23 = torch.zeros(Mp * D, **f64), torch.zeros(Mp * D, **f64)
        self.XA, self.XB = torch.zeros(D * D, **f64), torch.zeros(D * D, **f64)
        self.xvA, self.xvB = torch.zeros(D, **f64), torch.zeros(D, **f64)
        self.k1v, self.k23v = torch.zeros(Mp, **f64), torch.zeros(Mp, **f64)
        self.mid, self._mid_key = torch.zeros(Mp * D, **f64), None

    # ---------------------------------------------------------------------------------------------------------
    def _to_dev(self, a):
        import torch
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=self.device)

    def _stage(self, fwd, A, a0, a1, mat_a, X, xvec, base, base_off, vbase, vbase_off, E, e0_off, e1_off, ev, ev0, ev1,
               out, out_off, vout, vout_off, kstore, final_mode, cx, cf, J=None, j_off=0, jv=None, jv_off=0):
        """One RK stage for this rank's row block.  Offsets are in doubles into flat tensors.
        mat_a = (tensor, off0, off1_or_None): A operand of the matrix product (may differ from the vector's A: RK2)."""
        D, Mp, row0 = self.D, self.Mp, self.row0
        mt, m0, m1 = mat_a
        if m1 is not None:
            # mid-point operand 0.5 (A_k + A_{k+1}) of this rank's slab, formed once and shared by the two stages
            # that use it (a GEMM that averages while staging streams both operands: 2.3x slower at D = 1024)
            key = (mt.data_ptr(), m0, m1, bool(fwd))
            if self._mid_key != key:
                D2 = D * D
                if fwd:      # rows I_p: contiguous [Mp][D]
                    torch_add = mt[m0 + row0 * D:m0 + (row0 + Mp) * D] + mt[m1 + row0 * D:m1 + (row0 + Mp) * D]
                else:        # columns I_p of A (rows of A^T): [D][Mp], leading dimension Mp
                    torch_add = (mt[m0:m0 + D2].view(D, D)[:, row0:row0 + Mp] +
                                 mt[m1:m1 + D2].view(D, D)[:, row0:row0 + Mp]).reshape(-1)
                self.mid[:Mp * D] = torch_add * 0.5
                self._mid_key = key
            if fwd:
                self.backend.gemm(False, Mp, D, D, self.mid, 0, None, 0, D, X, D, self.Wp, self.cw)
            else:
                self.backend.gemm(True, Mp, D, D, self.mid, 0, None, 0, Mp, X, D, self.Wp, self.cw)
        elif fwd:    # W[I_p, :] = A[I_p, :] . X
            self.backend.gemm(False, Mp, D, D, mt, m0 + row0 * D, None, 0, D, X, D, self.Wp, self.cw)
        else:        # W'[I_p, :] = (A^T)[I_p, :] . Psi
            self.backend.gemm(True, Mp, D, D, mt, m0 + row0, None, 0, D, X, D, self.Wp, self.cw)
        if self.world > 1:
            self.dist.all_to_all_single(self.Wcol, self.Wp, group=self.group)
        self.backend.stage(D=D, row0=row0, Mp=Mp, cw=self.cw, fwd=int(fwd), kstore=kstore, final_mode=final_mode, lda=D,
                           cx=cx, cf=cf, W=self.Wp, Wcol=self.Wcol,
                           E0=(E, e0_off + row0 * D), E1=(E, e1_off + row0 * D) if e1_off is not None else None,
                           J=(J, j_off + row0 * D) if J is not None else None,
                           base=(base, base_off + row0 * D), K1=self.K1, K23=self.K23,
                           out=(out, out_off + row0 * D),
                           A0=(A, a0), A1=(A, a1) if a1 is not None else None, x=xvec,
                           e0=(ev, ev0 + row0), e1=(ev, ev1 + row0) if ev1 is not None else None,
                           jv=(jv, jv_off + row0) if jv is not None else None,
                           vbase=(vbase, vbase_off + row0), k1v=self.k1v, k23v=self.k23v,
                           vout=(vout, vout_off + row0))
        if self.world > 1:
            D2 = D * D
            full = out[out_off:out_off + D2]
            mine, vfull = full[row0 * D:(row0 + Mp) * D], vout[vout_off:vout_off + D]
            vmine = vfull[row0:row0 + Mp]
            if self.device.type == "cpu":      # gloo (tests): no in-place aliasing
                mine, vmine = mine.clone(), vmine.clone()
            self.dist.all_gather_into_tensor(full, mine, group=self.group)
            self.dist.all_gather_into_tensor(vfull, vmine, group=self.group)

    # ---------------------------------------------------------------------------------------------------------
    def solve_fwd(self, lin_a, off_b, m0, s0, sigma):
        import torch
        self._mid_key = None
        A, b = self._to_dev(lin_a).reshape(-1), self._to_dev(off_b).reshape(-1)
        S0, Sg = self._to_dev(s0), self._to_dev(sigma)
        if not (_is_symmetric(S0.reshape(self.D, self.D)) and _is_symmetric(Sg.reshape(self.D, self.D))):
            raise NotImplementedError("the large-D path needs symmetric s0 and sigma")
        D, D2, dt, h = self.D, self.D * self.D, self.dt, 0.5 * self.dt
        n = off_b.shape[0]
        S = torch.zeros(n * D2, dtype=torch.float64, device=self.device)
        m = torch.zeros(n * D, dtype=torch.float64, device=self.device)
        S[:D2] = S0.reshape(-1)
        m[:D] = self._to_dev(m0).reshape(-1)
        Sg = Sg.reshape(-1)
        XA, XB, xvA, xvB = self.XA, self.XB, self.xvA, self.xvB
        for k in range(n - 1):
            ak, ak1, bk, bk1 = k * D2, (k + 1) * D2, k * D, (k + 1) * D
            Sk, mk = S[ak:ak + D2], m[bk:bk + D]
            common = dict(base=S, base_off=ak, vbase=m, vbase_off=bk, E=Sg, e0_off=0, e1_off=None, ev=b)
            if self.method == "euler":
                self._stage(True, A, ak, None, (A, ak, None), Sk, mk, ev0=bk, ev1=None, out=S, out_off=ak1, vout=m,
                            vout_off=bk1, kstore=0, final_mode=1, cx=0.0, cf=dt, **common)
            elif self.method == "heun":
                self._stage(True, A, ak, None, (A, ak, None), Sk, mk, ev0=bk, ev1=None, out=XA, out_off=0, vout=xvA,
                            vout_off=0, kstore=1, final_mode=0, cx=dt, cf=0.0, **common)
                self._stage(True, A, ak1, None, (A, ak1, None), XA, xvA, ev0=bk1, ev1=None, out=S, out_off=ak1, vout=m,
                            vout_off=bk1, kstore=0, final_mode=2, cx=0.0, cf=h, **common)
            elif self.method == "rk2":
                # covariance predictor: S_k stands in for A_k (reference quirk, runge_kutta2.py:96); mean: A_k
                self._stage(True, A, ak, None, (S, ak, None), Sk, mk, ev0=bk, ev1=None, out=XA, out_off=0, vout=xvA,
                            vout_off=0, kstore=0, final_mode=0, cx=h, cf=0.0, **common)
                self._stage(True, A, ak, ak1, (A, ak, ak1), XA, xvA, ev0=bk1, ev1=bk, out=S, out_off=ak1, vout=m,
                            vout_off=bk1, kstore=0, final_mode=1, cx=0.0, cf=dt, **common)
            else:
                self._stage(True, A, ak, None, (A, ak, None), Sk, mk, ev0=bk, ev1=None, out=XA, out_off=0, vout=xvA,
                            vout_off=0, kstore=1, final_mode=0, cx=h, cf=0.0, **common)
                self._stage(True, A, ak, ak1, (A, ak, ak1), XA, xvA, ev0=bk1, ev1=bk, out=XB, out_off=0, vout=xvB,
                            vout_off=0, kstore=2, final_mode=0, cx=h, cf=0.0, **common)
                self._stage(True, A, ak, ak1, (A, ak, ak1), XB, xvB, ev0=bk1, ev1=bk, out=XA, out_off=0, vout=xvA,
                            vout_off=0, kstore=3, final_mode=0, cx=dt, cf=0.0, **common)
                self._stage(True, A, ak1, None, (A, ak1, None), XA, xvA, ev0=bk1, ev1=None, out=S, out_off=ak1, vout=m,
                            vout_off=bk1, kstore=0, final_mode=3, cx=0.0, cf=dt, **common)
        return m.reshape(n, D), S.reshape(n, D, D)

    def solve_bwd(self, lin_a, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds):
        import torch
        self._mid_key = None
        A = self._to_dev(lin_a).reshape(-1)
        gm, gs = self._to_dev(dEsde_dm).reshape(-1), self._to_dev(dEsde_ds)
        jm, js = self._to_dev(dEobs_dm).reshape(-1), self._to_dev(dEobs_ds)
        D, D2, dt, h = self.D, self.D * self.D, self.dt, 0.5 * self.dt
        if not (_is_symmetric(gs.reshape(-1, D, D)) and _is_symmetric(js.reshape(-1, D, D))):
            raise NotImplementedError("the large-D path needs symmetric dEsde_ds / dEobs_ds")
        gs, js = gs.reshape(-1), js.reshape(-1)
        n = dEsde_dm.shape[0]
        psi = torch.zeros(n * D2, dtype=torch.float64, device=self.device)
        lam = torch.zeros(n * D, dtype=torch.float64, device=self.device)
        XA, XB, xvA, xvB = self.XA, self.XB, self.xvA, self.xvB
        for t in range(n - 1, 0, -1):
            at, am, vt, vm = t * D2, (t - 1) * D2, t * D, (t - 1) * D
            Pt, lt = psi[at:at + D2], lam[vt:vt + D]
            common = dict(base=psi, base_off=at, vbase=lam, vbase_off=vt, E=gs, ev=gm)
            fin = dict(out=psi, out_off=am, vout=lam, vout_off=vm, J=js, j_off=am, jv=jm, jv_off=vm)
            if self.method == "euler":
                self._stage(False, A, at, None, (A, at, None), Pt, lt, e0_off=at, e1_off=None, ev0=vt, ev1=None,
                            kstore=0, final_mode=1, cx=0.0, cf=dt, **common, **fin)
            elif self.method == "heun":
                self._stage(False, A, at, None, (A, at, None), Pt, lt, e0_off=at, e1_off=None, ev0=vt, ev1=None,
                            out=XA, out_off=0, vout=xvA, vout_off=0, kstore=1, final_mode=0, cx=dt, cf=0.0, **common)
                self._stage(False, A, am, None, (A, am, None), XA, xvA, e0_off=am, e1_off=None, ev0=vm, ev1=None,
                            kstore=0, final_mode=2, cx=0.0, cf=h, **common, **fin)
            elif self.method == "rk2":
                self._stage(False, A, at, None, (A, at, None), Pt, lt, e0_off=at, e1_off=None, ev0=vt, ev1=None,
                            out=XA, out_off=0, vout=xvA, vout_off=0, kstore=0, final_mode=0, cx=h, cf=0.0, **common)
                self._stage(False, A, am, at, (A, am, at), XA, xvA, e0_off=at, e1_off=am, ev0=vt, ev1=vm,
                            kstore=0, final_mode=1, cx=0.0, cf=dt, **common, **fin)
            else:
                self._stage(False, A, at, None, (A, at, None), Pt, lt, e0_off=at, e1_off=None, ev0=vt, ev1=None,
                            out=XA, out_off=0, vout=xvA, vout_off=0, kstore=1, final_mode=0, cx=h, cf=0.0, **common)
                self._stage(False, A, am, at, (A, am, at), XA, xvA, e0_off=at, e1_off=am, ev0=vt, ev1=vm,
                            out=XB, out_off=0, vout=xvB, vout_off=0, kstore=2, final_mode=0, cx=h, cf=0.0, **common)
                self._stage(False, A, am, at, (A, am, at), XB, xvB, e0_off=at, e1_off=am, ev0=vt, ev1=vm,
                            out=XA, out_off=0, vout=xvA, vout_off=0, kstore=3, final_mode=0, cx=dt, cf=0.0, **common)
                self._stage(False, A, am, None, (A, am, None), XA, xvA, e0_off=am, e1_off=None, ev0=vm, ev1=None,
                            kstore=0, final_mode=3, cx=0.0, cf=dt, **common, **fin)
        return lam.reshape(n, D), psi.reshape(n, D, D)


