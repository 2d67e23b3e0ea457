"""
Large-D (D > 64) recursions ROW-SHARDED over the GPUs of one node: the host side of `vgpa_shard_*`
(vgpa_amd/csrc/large_d.hip), whose C++ driver runs the whole step / stage loop of a sweep AND its collectives:

    per RK stage, on rank p owning rows I_p = [row0, row0 + Mp):
        W[I_p, :]   = A_s[I_p, :] . X            fp64-MFMA GEMM, K-chunk launches behind the sub-blocks of the previous gather
        Wcol        = all_to_all(W[I_p, :])      -> W[:, I_p]  (RCCL over xGMI)
        X'[I_p, :]  = S_k[I_p, :] + c * ( -(W[I_p, :] + Wcol^T) + Sigma[I_p, :] )     fused element-wise kernel
        X'          = gather(X'[I_p, :])         pipelined sub-blocks on a second stream (RCCL point to point)

Same steppers and quirks as the reference (src/numerics/{euler,heun,runge_kutta2,runge_kutta4}.py).

Device memory comes from the library (`_lib.DeviceArray`: vgpa_device_alloc / _memcpy) and the work runs on the shard's own
streams: this module needs no tensor library.  torch appears in two places only, both plumbing: `torch.distributed` hands the
RCCL unique id to the ranks (any backend) when no communicator table is injected, and `HostStagedComm` -- the table that lets
real processes exercise the driver where RCCL cannot run (several ranks on one GPU) -- stages through torch.distributed.
Callers that already hold torch tensors may pass them (anything with `data_ptr()` is consumed in place).
"""
import ctypes
import os
import sys

import numpy as np

from ._lib import load, _raise, DeviceArray

METHODS = ("euler", "heun", "rk2", "rk4")


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


class ShardProblem:
    """What `NativeShardedRecursion.prepare` returns: the vgpa_shard_problem struct of one variational problem and the device arrays
    its pointers refer to (kept alive with it)."""

    def __init__(self, owner, prob, keep):
        self.owner, self.prob, self.keep = owner, prob, keep


class NativeShardedRecursion:
    """
    The row-sharded recursion with the whole step / stage loop AND its collectives inside libvgpa_hip.so
    (vgpa_shard_solve_fwd / _bwd, vgpa_amd/csrc/large_d.hip): per RK stage one all-to-all of the packed product blocks and
    one grouped all-gather (matrix row blocks + vector entries) over RCCL, enqueued on the shard's stream between the
    kernels; no Python and no host synchronisation inside a sweep.  The results are TIME-sharded: a rank keeps only the
    grid points [t_lo, t_hi) it owns (`time_slice`), which is the layout the time-parallel energy / gradient phase uses.

    `comm`: a `_lib.VgpaComm` (tests inject one built from Python callbacks); by default RCCL, whose unique id rank 0
    creates and torch.distributed (any backend, e.g. gloo) broadcasts.  One rank: no communicator at all.
    Inputs: host arrays (uploaded into library-owned device memory) or anything with `data_ptr()` (torch tensors, `DeviceArray`:
    consumed in place); outputs: `_lib.DeviceArray` (`.numpy()`, `.cpu().numpy()`, `__cuda_array_interface__`).
    """

    def __init__(self, method, dt, dim_d, n_pts, rank=None, world=None, device=None, comm=None):
        from ._lib import VgpaComm, METHOD_IDS
        dist = None
        if "torch" in sys.modules:               # (a process group can only exist where torch has been imported)
            import torch.distributed as dist
        method = str(method).lower()
        if method not in METHODS:
            raise ValueError(f" Integration method is unknown -> {method}.")
        if dt <= 0.0:
            raise ValueError(f" Discrete time step should be strictly positive -> {dt}.")
        self._lib = load()
        ready = dist is not None and dist.is_available() and dist.is_initialized()
        self.world = int(world if world is not None else (dist.get_world_size() if ready else 1))
        self.rank = int(rank if rank is not None else (dist.get_rank() if ready else 0))
        self.D, self.Np = int(dim_d), int(n_pts)
        if self.D % self.world:
            raise ValueError(f"D={self.D} must be a multiple of the number of ranks ({self.world})")
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        self._comm, self._own_comm = None, False
        if self.world > 1:
            if comm is None:
                import torch                     # rendezvous plumbing only: the unique id travels through torch.distributed
                import torch.distributed as dist
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
        """Device memory for an operand: host arrays go up into a library-owned buffer; anything that already lives on the device
        (`data_ptr()`: a DeviceArray, a torch tensor) is used where it is -- fp64, contiguous, on this rank's device."""
        if hasattr(a, "data_ptr"):
            if "torch" in sys.modules:
                import torch
                if isinstance(a, torch.Tensor):
                    a = a.to(device=torch.device("cuda", self.device), dtype=torch.float64).contiguous()
            return a
        return DeviceArray.from_host(np.ascontiguousarray(a, dtype=np.float64), self.device)

    def _external_ready(self):
        """Operands handed over as tensors of a library with its own streams must be complete before the shard's streams read."""
        if "torch" in sys.modules:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize(self.device)

    def _run(self, fn, inputs, what):
        ins = [self._dev(a) for a in inputs]
        self._external_ready()
        n_own = self.time_slice[1] - self.time_slice[0]
        v_own = DeviceArray((max(n_own, 1), self.D), self.device)
        m_own = DeviceArray((max(n_own, 1), self.D, self.D), self.device)
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

    @property
    def rccl_communicators(self):
        """Communicators behind the RCCL table: 2 (the collectives of the compute stream and the point-to-point groups of the
        communication stream each have their own), 1 when librccl cannot split, 0 without an RCCL communicator."""
        if self._comm is None or not self._own_comm:
            return 0
        n = ctypes.c_int(0)
        self._check(self._lib.vgpa_rccl_comm_streams(ctypes.byref(self._comm), ctypes.byref(n)), "vgpa_rccl_comm_streams")
        return int(n.value)

    def prepare(self, theta, sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0):
        """The operands of `sweep` / `sweep_sharded` that do not change between the sweeps of one optimisation (the prior, the
        noises, the observations: two D x D uploads among them), uploaded ONCE: pass the returned object in place of `theta` and
        leave the other seven arguments out."""
        return ShardProblem(self, *self._problem(theta, sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0))

    def _problem_of(self, theta, rest):
        if isinstance(theta, ShardProblem):
            if theta.owner is not self:
                raise ValueError("this ShardProblem was prepared by another recursion")
            return theta.prob, theta.keep
        if any(v is None for v in rest[:6]):
            raise TypeError("sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag are required unless a prepared ShardProblem is passed")
        return self._problem(theta, *rest)

    def _gradient_arrays(self, out):
        """(gLa_own, gLb_own): new device arrays, or the caller's `out` pair from an earlier sweep (an optimisation keeps ONE pair:
        at D = 4096 a rank's gLa_own is gigabytes, and allocating / freeing it per sweep is host time the GPU cannot hide)."""
        d, n_own = self.D, max(self.time_slice[1] - self.time_slice[0], 1)
        if out is None:
            return DeviceArray((n_own, d, d), self.device), DeviceArray((n_own, d), self.device)
        ga, gb = out
        for arr, shape in ((ga, (d, d)), (gb, (d,))):
            if not isinstance(arr, DeviceArray) or arr.device != self.device or tuple(arr.shape[1:]) != shape:
                raise ValueError("out: the (gLa_own, gLb_own) pair an earlier sweep of this recursion returned is expected")
        ga, gb = (ga._base or ga), (gb._base or gb)         # (sweeps return [:n_own] views of the arrays they own)
        if ga.shape[0] != n_own or gb.shape[0] != n_own:
            raise ValueError("out: the (gLa_own, gLb_own) pair an earlier sweep of this recursion returned is expected")
        return ga, gb

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

    def sweep_sharded(self, a_own, b_own, theta, sigma_diag=None, m0=None, s0=None, obs_t=None, obs_y=None, obs_noise_diag=None,
                      e0=0.0, *, out=None):
        """
        vgpa_shard_sweep_sharded: the fused sweep with x MEMORY-SHARDED like the gradient -- a_own [n_own, D, D] and b_own
        [n_own, D] are A_t / b_t of the grid points of `time_slice` only (device tensors or host arrays).  Returns
        (F, gLa_own, gLb_own) like `sweep`.  No rank holds a complete (Np, D, D) array.  `theta` may be a `prepare`d problem
        (then nothing is uploaded but a_own / b_own), `out` the gradient pair of an earlier sweep (written again, returned again).
        """
        d = self.D
        n_own = self.time_slice[1] - self.time_slice[0]
        ad, bd = self._dev(a_own), self._dev(b_own)
        if n_own == 0:
            ad, bd = self._dev(np.zeros((1, d, d))), self._dev(np.zeros((1, d)))
        prob, keep = self._problem_of(theta, (sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0))
        ga, gb = self._gradient_arrays(out)
        f = ctypes.c_double(0.0)
        self._external_ready()
        self._check(self._lib.vgpa_shard_sweep_sharded(self._h, ctypes.byref(prob), ctypes.c_void_p(ad.data_ptr()),
                                                       ctypes.c_void_p(bd.data_ptr()), ctypes.byref(f),
                                                       ctypes.c_void_p(ga.data_ptr()), ctypes.c_void_p(gb.data_ptr())),
                    "vgpa_shard_sweep_sharded")
        del keep
        return f.value, ga[:n_own], gb[:n_own]

    def sweep(self, x, theta, sigma_diag=None, m0=None, s0=None, obs_t=None, obs_y=None, obs_noise_diag=None, e0=0.0, *, out=None):
        """
        Free energy and gradient of VarGP (variational.py:141-288) for ONE Lorenz-96 problem on the row-sharded recursion
        (vgpa_shard_sweep): F on every rank, the gradient TIME-sharded -- returns (F, gLa_own [n_own, D, D], gLb_own [n_own, D])
        as device tensors for the grid points of `time_slice`.  x = [A_t | b_t] (host array or device tensor, replicated);
        diagonal system noise `sigma_diag`, diagonal observation noise `obs_noise_diag`, identity observation operator.
        The outcome is collective: every rank raises the same error (LinAlgError when S_t of ANY rank's grid points is not
        positive definite).  `theta` may be a `prepare`d problem, `out` the gradient pair of an earlier sweep (see `sweep_sharded`).
        """
        d = self.D
        xd = self._dev(x)
        prob, keep = self._problem_of(theta, (sigma_diag, m0, s0, obs_t, obs_y, obs_noise_diag, e0))
        n_own = self.time_slice[1] - self.time_slice[0]
        ga, gb = self._gradient_arrays(out)
        f = ctypes.c_double(0.0)
        self._external_ready()
        self._check(self._lib.vgpa_shard_sweep(self._h, ctypes.byref(prob), ctypes.c_void_p(xd.data_ptr()), ctypes.byref(f),
                                               ctypes.c_void_p(ga.data_ptr()), ctypes.c_void_p(gb.data_ptr())), "vgpa_shard_sweep")
        del keep
        return f.value, ga[:n_own], gb[:n_own]
