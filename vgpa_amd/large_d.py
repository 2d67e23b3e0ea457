"""
Large-D (D > 64) forward / backward recursions: host driver of the stage kernels `vgpa_ld_gemm` / `vgpa_ld_stage`
(vgpa_amd/csrc/large_d.hip), optionally ROW-SHARDED over the GPUs of one node:

    per RK stage, on rank p owning rows I_p = [row0, row0 + Mp):
        W[I_p, :]   = A_s[I_p, :] . X            fp64-MFMA GEMM, written in column-chunk packed layout
        Wcol        = all_to_all(W[I_p, :])      -> W[:, I_p]  (RCCL over xGMI; identity with one rank)
        X'[I_p, :]  = S_k[I_p, :] + c * ( -(W[I_p, :] + Wcol^T) + Sigma[I_p, :] )     fused element-wise kernel
        X'          = all_gather(X'[I_p, :])     (RCCL; identity with one rank)

Same steppers and quirks as the reference (src/numerics/{euler,heun,runge_kutta2,runge_kutta4}.py); the symmetric
shortcut W + W^T needs symmetric S0 / Sigma (forward) and symmetric dEsde_dS / jumps (backward) -- checked.

torch is plumbing here: device memory, the current HIP stream and torch.distributed (backend "nccl" = RCCL).  The two
stage operations are reached through a small `backend` object; the package ships exactly one, `HipStageBackend`
(the HIP kernels).  Tests inject a CPU stand-in to exercise the sharding / collective logic under gloo.
"""
import ctypes

import numpy as np

from ._lib import load, LdStageArgs, _raise
from .parallel import shard_range

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


# ---------------------------------------------------------------------------------------------------------------------
def _hip_runtime():
    """ctypes handle of the HIP runtime this process already uses (torch bundles its own copy: a second one would not know
    the streams of the first)."""
    import torch  # noqa: F401
    load()
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    if not paths:
        raise RuntimeError("HIP runtime not loaded")
    return ctypes.CDLL(paths[0])


class HostStagedComm:
    """
    A `vgpa_comm` table on ANY torch.distributed backend (gloo in particular), with the data staged through the host: device ->
    host copy, CPU collective, host -> device copy, every call synchronous.  Orders of magnitude slower than RCCL over xGMI and
    never used for a measurement -- it exists so that the native row-sharded driver (`vgpa_shard_*`: one PROCESS per rank) can be
    exercised with real processes where RCCL cannot be (several ranks on one GPU: tests, `VGPA_BENCH_REHEARSE=1`).  Point-to-point
    groups map to `batch_isend_irecv`; `abort` marks the table dead (every later call fails) -- the peers leave through their own
    time-outs, as with a dead RCCL peer.
    """

    def __init__(self, group=None):
        import torch.distributed as dist
        from ._lib import VgpaComm, COMM_COLLECTIVE, COMM_GROUP, COMM_P2P
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.hip = _hip_runtime()
        self.dead = False
        self._ops = None
        t = VgpaComm()
        self._keep = (COMM_COLLECTIVE(self._guard(self._all_gather)), COMM_COLLECTIVE(self._guard(self._all_to_all)),
                      COMM_GROUP(self._guard(self._group_begin)), COMM_GROUP(self._guard(self._group_end)),
                      COMM_P2P(self._guard(self._send)), COMM_P2P(self._guard(self._recv)), COMM_GROUP(self._abort))
        t.user, t.all_gather, t.all_to_all, t.group_begin, t.group_end, t.send, t.recv, t.abort = (None, *self._keep)
        self.table = t

    def _guard(self, fn):
        def call(*a):
            if self.dead:
                return 5
            try:
                return fn(*a)
            except BaseException:            # noqa: BLE001 - nothing may unwind through the C caller
                return 9
        return call

    def _to_host(self, ptr, count, stream):
        import torch
        self.hip.hipStreamSynchronize(ctypes.c_void_p(stream))
        h = torch.empty(int(count), dtype=torch.float64)
        if count:
            self.hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(8 * int(count)), 2)
        return h

    def _to_device(self, ptr, h):
        if h.numel():
            self.hip.hipMemcpy(ctypes.c_void_p(ptr), ctypes.c_void_p(h.data_ptr()), ctypes.c_size_t(8 * h.numel()), 1)

    def _all_gather(self, user, send, recv, count, stream):
        import torch
        mine = self._to_host(send, count, stream)
        out = [torch.empty(int(count), dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(out, mine, group=self.group)
        for q in range(self.world):
            self._to_device(recv + 8 * q * count, out[q])
        return 0

    def _all_to_all(self, user, send, recv, count, stream):
        import torch
        mine = self._to_host(send, count * self.world, stream)
        out = [torch.empty(int(count) * self.world, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(out, mine, group=self.group)         # (gloo has no all-to-all of its own on every build)
        for q in range(self.world):
            self._to_device(recv + 8 * q * count, out[q][self.rank * count:(self.rank + 1) * count].contiguous())
        return 0

    def _group_begin(self, user):
        self._ops = []
        return 0

    def _send(self, user, buf, count, peer, stream):
        self._ops.append(("s", buf, int(count), int(peer), stream))
        return 0

    def _recv(self, user, buf, count, peer, stream):
        self._ops.append(("r", buf, int(count), int(peer), stream))
        return 0

    def _group_end(self, user):
        import torch
        ops, self._ops = self._ops or [], None
        if not ops:
            return 0
        p2p, landing = [], []
        for kind, buf, count, peer, stream in ops:
            gpeer = peer if self.group is None else self.dist.get_global_rank(self.group, peer)
            if kind == "s":
                p2p.append(self.dist.P2POp(self.dist.isend, self._to_host(buf, count, stream), gpeer, group=self.group))
            else:
                h = torch.empty(count, dtype=torch.float64)
                landing.append((buf, h))
                p2p.append(self.dist.P2POp(self.dist.irecv, h, gpeer, group=self.group))
        for w in self.dist.batch_isend_irecv(p2p):
            w.wait()
        for buf, h in landing:
            self._to_device(buf, h)
        return 0

    def _abort(self, user):
        self.dead = True
        return 0


class NativeShardedRecursion:
    """
    The row-sharded recursion with the whole step / stage loop AND its collectives inside libvgpa_hip.so
    (vgpa_shard_solve_fwd / _bwd, vgpa_amd/csrc/large_d.hip): per RK stage one all-to-all of the packed product blocks and
    one grouped all-gather (matrix row blocks + vector entries) over RCCL, enqueued on the shard's stream between the
    kernels; no Python and no host synchronisation inside a sweep.  The results are TIME-sharded: a rank keeps only the
    grid points [t_lo, t_hi) it owns (`time_slice`), which is the layout the time-parallel energy / gradient phase uses.

    `comm`: a `_lib.VgpaComm` (tests inject one built from Python callbacks); by default RCCL, whose unique id rank 0
    creates and torch.distributed (any backend, e.g. gloo) broadcasts.  One rank: no communicator at all.
    `ShardedRecursion` above drives the same two kernels stage by stage from Python; it is kept as the CPU / gloo-testable
    statement of the schedule.
    """

    def __init__(self, method, dt, dim_d, n_pts, rank=None, world=None, device=None, comm=None):
        import torch
        import torch.distributed as dist
        from ._lib import VgpaComm, METHOD_IDS
        method = str(method).lower()
        if method not in METHODS:
            raise ValueError(f" Integration method is unknown -> {method}.")
        if dt <= 0.0:
            raise ValueError(f" Discrete time step should be strictly positive -> {dt}.")
        self._lib = load()
        ready = dist.is_available() and dist.is_initialized()
        self.world = int(world if world is not None else (dist.get_world_size() if ready else 1))
        self.rank = int(rank if rank is not None else (dist.get_rank() if ready else 0))
        self.D, self.Np = int(dim_d), int(n_pts)
        if self.D % self.world:
            raise ValueError(f"D={self.D} must be a multiple of the number of ranks ({self.world})")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._comm, self._own_comm = None, False
        if self.world > 1:
            if comm is None:
                comm = VgpaComm()
                uid = torch.zeros(128, dtype=torch.uint8)
                if self.rank == 0:
                    buf = (ctypes.c_char * 128)()
                    self._check(self._lib.vgpa_rccl_unique_id(buf), "vgpa_rccl_unique_id")
                    uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
                if dist.get_backend() == "nccl":
                    uid = uid.cuda(self.device)
                dist.broadcast(uid, src=0)
                raw = bytes(uid.cpu().numpy().tobytes())
                self._check(self._lib.vgpa_rccl_comm_create(ctypes.byref(comm), raw, self.rank, self.world, self.device),
                            "vgpa_rccl_comm_create")
                self._own_comm = True
            self._comm = comm
        h = ctypes.c_void_p()
        self._check(self._lib.vgpa_shard_create(ctypes.byref(h), METHOD_IDS[method], float(dt), self.D, self.Np, self.rank,
                                                self.world, self.device,
                                                ctypes.byref(self._comm) if self._comm is not None else None, None),
                    "vgpa_shard_create")
        self._h = h
        lo, hi = ctypes.c_int(), ctypes.c_int()
        self._lib.vgpa_shard_time_slice(self._h, ctypes.byref(lo), ctypes.byref(hi))
        self.time_slice = (lo.value, hi.value)

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            _raise(rc, f"{what} failed")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vgpa_shard_destroy(self._h)
            self._h = None
        if self._own_comm and self._comm is not None:
            self._lib.vgpa_rccl_comm_destroy(ctypes.byref(self._comm))
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._lib.vgpa_shard_synchronize(self._h), "vgpa_shard_synchronize")

    def _dev(self, a):
        import torch
        dev = torch.device("cuda", self.device)
        if isinstance(a, torch.Tensor):
            return a.to(device=dev, dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)

    def _run(self, fn, inputs, what):
        import torch
        ins = [self._dev(a) for a in inputs]
        torch.cuda.synchronize(self.device)                       # uploads (torch's stream) before the shard's stream reads
        n_own = self.time_slice[1] - self.time_slice[0]
        dev = torch.device("cuda", self.device)
        v_own = torch.empty((max(n_own, 1), self.D), dtype=torch.float64, device=dev)
        m_own = torch.empty((max(n_own, 1), self.D, self.D), dtype=torch.float64, device=dev)
        ptr = [ctypes.c_void_p(t.data_ptr()) for t in ins]
        self._check(fn(self._h, *ptr, ctypes.c_void_p(v_own.data_ptr()), ctypes.c_void_p(m_own.data_ptr())), what)
        self.synchronize()
        return v_own[:n_own], m_own[:n_own]

    def solve_fwd(self, lin_a, off_b, m0, s0, sigma):
        """(m_t, S_t) for the grid points of `time_slice`."""
        return self._run(self._lib.vgpa_shard_solve_fwd, (lin_a, off_b, m0, s0, sigma), "vgpa_shard_solve_fwd")

    def solve_bwd(self, lin_a, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds):
        """(lam_t, Psi_t) for the grid points of `time_slice`."""
        return self._run(self._lib.vgpa_shard_solve_bwd, (lin_a, dEsde_dm, dEsde_ds, dEobs_dm, dEobs_ds),
                         "vgpa_shard_solve_bwd")

    def set_option(self, option, value):
        """vgpa_shard_set_option: SHARD_OPT_GATHER_CHUNKS (0 = serial schedule, 1..8 sub-blocks of the pipelined gather; the same
        value on every rank), SHARD_OPT_TIMEOUT_MS."""
        self._check(self._lib.vgpa_shard_set_option(self._h, int(option), int(value)), "vgpa_shard_set_option")

    @property
    def gather_chunks(self):
        """Sub-blocks of the pipelined gather actually in use (0: the serial schedule)."""
        from ._lib import SHARD_OPT_GATHER_CHUNKS
        v = ctypes.c_int64(0)
        self._check(self._lib.vgpa_shard_get_option(self._h, SHARD_OPT_GATHER_CHUNKS, ctypes.byref(v)), "vgpa_shard_get_option")
        return int(v.value)

    def time_collectives(self, reps=20):
        """(all_to_all_ms, gather_ms): the two per-stage collectives alone, as a stage issues them (collective call)."""
        a, g = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._check(self._lib.vgpa_shard_time_collectives(self._h, int(reps), ctypes.byref(a), ctypes.byref(g)),
                    "vgpa_shard_time_collectives")
        return a.value, g.value

    def time_stage(self, reps=20, with_collectives=True):
        """Milliseconds of ONE stage of the forward RK4 recursion as the driver issues it (vgpa_shard_time_stage); without the
        collectives it is the stage's compute alone.  Timing only: overwrites the shard's workspace."""
        ms = ctypes.c_double(0.0)
        self._check(self._lib.vgpa_shard_time_stage(self._h, int(reps), 1 if with_collectives else 0, ctypes.byref(ms)),
                    "vgpa_shard_time_stage")
        return ms.value

    def phase_ms(self):
        """Phases of this rank's last fused sweep in milliseconds (vgpa_shard_phase_ms)."""
        v = (ctypes.c_double * 6)()
        self._check(self._lib.vgpa_shard_phase_ms(self._h, v), "vgpa_shard_phase_ms")
        names = ("x_exchange", "forward_recursion", "energy_obs", "exchanges", "backward_recursion", "gradient")
        return {k: float(x) for k, x in zip(names, v)}

    @property
    def rccl_ranks(self):
        """Ranks of the communicator as librccl itself counts them (ncclCommCount); 0 without an RCCL communicator (one rank,
        or an injected table)."""
        if self._comm is None or not self._own_comm:
            return 0
        n = ctypes.c_int(0)
        self._check(self._lib.vgpa_rccl_comm_count(ctypes.byref(self._comm), ctypes.byref(n)), "vgpa_rccl_comm_count")
        return int(n.value)

    def _problem(self, theta, sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0):
        from ._lib import VgpaShardProblem
        d = self.D
        sig = np.asarray(sigma_diag, dtype=np.float64).reshape(d)
        rdiag = np.asarray(obs_noise_diag, dtype=np.float64).reshape(d)
        obs_t = np.ascontiguousarray(obs_t, dtype=np.int64)
        m_obs = int(obs_t.size)
        keep = [obs_t, self._dev(1.0 / sig), self._dev(m0), self._dev(np.asarray(s0, dtype=np.float64).reshape(d, d)),
                self._dev(np.diag(sig)), self._dev(np.asarray(obs_y, dtype=np.float64).reshape(max(m_obs, 1), d) if m_obs else np.zeros((1, d))),
                self._dev(1.0 / rdiag)]
        prob = VgpaShardProblem()
        prob.theta = float(theta)
        prob.inv_sigma_diag, prob.m0, prob.s0, prob.sigma = (keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr())
        prob.n_obs = m_obs
        prob.obs_t = obs_t.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        prob.obs_y, prob.obs_rinv_diag = keep[5].data_ptr(), keep[6].data_ptr()
        prob.obs_const = m_obs * (d * np.log(2.0 * np.pi) + float(np.sum(np.log(rdiag))))
        prob.e0 = float(e0)
        return prob, keep

    def sweep_sharded(self, a_own, b_own, theta, sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0):
        """
        vgpa_shard_sweep_sharded: the fused sweep with x MEMORY-SHARDED like the gradient -- a_own [n_own, D, D] and b_own
        [n_own, D] are A_t / b_t of the grid points of `time_slice` only (device tensors or host arrays).  Returns
        (F, gLa_own, gLb_own) like `sweep`.  No rank holds a complete (Np, D, D) array.
        """
        import torch
        d = self.D
        n_own = self.time_slice[1] - self.time_slice[0]
        ad, bd = self._dev(a_own), self._dev(b_own)
        if n_own == 0:
            ad, bd = self._dev(np.zeros((1, d, d))), self._dev(np.zeros((1, d)))
        prob, keep = self._problem(theta, sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0)
        dev = torch.device("cuda", self.device)
        ga = torch.empty((max(n_own, 1), d, d), dtype=torch.float64, device=dev)
        gb = torch.empty((max(n_own, 1), d), dtype=torch.float64, device=dev)
        f = ctypes.c_double(0.0)
        torch.cuda.synchronize(self.device)
        self._check(self._lib.vgpa_shard_sweep_sharded(self._h, ctypes.byref(prob), ctypes.c_void_p(ad.data_ptr()),
                                                       ctypes.c_void_p(bd.data_ptr()), ctypes.byref(f),
                                                       ctypes.c_void_p(ga.data_ptr()), ctypes.c_void_p(gb.data_ptr())),
                    "vgpa_shard_sweep_sharded")
        del keep
        return f.value, ga[:n_own], gb[:n_own]

    def sweep(self, x, theta, sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0):
        """
        Free energy and gradient of VarGP (variational.py:141-288) for ONE Lorenz-96 problem on the row-sharded recursion
        (vgpa_shard_sweep): F on every rank, the gradient TIME-sharded -- returns (F, gLa_own [n_own, D, D], gLb_own [n_own, D])
        as device tensors for the grid points of `time_slice`.  x = [A_t | b_t] (host array or device tensor, replicated);
        diagonal system noise `sigma_diag`, diagonal observation noise `obs_noise_diag`, identity observation operator.
        The outcome is collective: every rank raises the same error (LinAlgError when S_t of ANY rank's grid points is not
        positive definite).
        """
        import torch
        d = self.D
        xd = self._dev(x)
        prob, keep = self._problem(theta, sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0)
        n_own = self.time_slice[1] - self.time_slice[0]
        dev = torch.device("cuda", self.device)
        ga = torch.empty((max(n_own, 1), d, d), dtype=torch.float64, device=dev)
        gb = torch.empty((max(n_own, 1), d), dtype=torch.float64, device=dev)
        f = ctypes.c_double(0.0)
        torch.cuda.synchronize(self.device)
        self._check(self._lib.vgpa_shard_sweep(self._h, ctypes.byref(prob), ctypes.c_void_p(xd.data_ptr()), ctypes.byref(f),
                                               ctypes.c_void_p(ga.data_ptr()), ctypes.c_void_p(gb.data_ptr())), "vgpa_shard_sweep")
        del keep
        return f.value, ga[:n_own], gb[:n_own]
