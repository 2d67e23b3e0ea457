"""
ctypes binding of libvgpa_hip.so (include/vgpa_hip.h).  Thin: argument marshalling, error mapping,
and a `Context` object that owns one `vgpa_ctx*`.

There is no CPU fallback here or anywhere in this package: if the shared library is missing, or no
HIP device is visible, the calls raise.
"""
import os
import ctypes
from ctypes import c_int, c_int32, c_int64, c_uint64, c_double, c_void_p, c_char_p, POINTER, byref

import numpy as np

ABI_VERSION = 2
ABI_DIAGNOSTIC_BUILD = 0x10000     # vgpa_hip.h: bit of vgpa_abi_version() set by a -DVGPA_EXPERIMENTS build
# VGPA_LIB: another build of the same library (same-box A/B of build variants: tools/build_variant.sh); never set by the package
LIB_PATH = os.environ.get("VGPA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvgpa_hip.so")

MODEL_IDS = {"NONE": -1, "OU": 0, "DW": 1, "L63": 2, "L96": 3}
METHOD_IDS = {"euler": 0, "heun": 1, "rk2": 2, "rk4": 3}
FETCH_IDS = {"mt": 0, "st": 1, "lamt": 2, "psit": 3, "Efx": 4, "Edf": 5, "dEsde_dm": 6, "dEsde_ds": 7, "Esde_t": 8}
FLAG_FORCE_GENERIC = 1
FLAG_STREAM_LARGE_D = 4
FLAG_LIBRARY_GEMM = 8
FLAG_SYM_UNITS = 16
FLAG_KEEP_PSI = 32
FLAG_MATERIALIZE = 64
OPT_LD_CHUNK = 1

# exported symbols, checked by the CPU test-suite against include/vgpa_hip.h
SYMBOLS = ["vgpa_create", "vgpa_destroy", "vgpa_last_error", "vgpa_abi_version", "vgpa_device_count",
           "vgpa_synchronize", "vgpa_stream", "vgpa_solve_fwd", "vgpa_solve_bwd", "vgpa_energy",
           "vgpa_obs_energy", "vgpa_free_energy", "vgpa_gradient", "vgpa_sweep", "vgpa_energy_parts",
           "vgpa_fetch", "vgpa_sweep_dev", "vgpa_free_energy_dev", "vgpa_sweep_enqueue", "vgpa_fetch_f",
           "vgpa_dev_alloc", "vgpa_dev_free", "vgpa_memcpy_h2d", "vgpa_memcpy_d2h",
           "vgpa_profile_begin", "vgpa_profile_end", "vgpa_ld_gemm", "vgpa_ld_stage", "vgpa_gradient_dev", "vgpa_energy_full", "vgpa_set_option", "vgpa_is_streaming", "vgpa_set_prior_energy",
           "vgpa_vec_dot", "vgpa_vec_absmax", "vgpa_vec_asum", "vgpa_vec_axpby", "vgpa_release_x",
           "vgpa_shard_create", "vgpa_shard_destroy", "vgpa_shard_time_slice", "vgpa_shard_stream", "vgpa_shard_synchronize",
           "vgpa_shard_solve_fwd", "vgpa_shard_solve_bwd", "vgpa_shard_sweep", "vgpa_shard_sweep_sharded", "vgpa_shard_set_option",
           "vgpa_shard_get_option", "vgpa_shard_time_collectives", "vgpa_ld_gemm_chunk", "vgpa_rccl_unique_id", "vgpa_rccl_comm_create",
           "vgpa_rccl_comm_destroy", "vgpa_rccl_comm_count", "vgpa_rccl_comm_streams", "vgpa_shard_time_stage", "vgpa_shard_phase_ms", "vgpa_time_slice",
           "vgpa_device_alloc", "vgpa_device_free", "vgpa_device_memcpy"]

P_DOUBLE = POINTER(c_double)


class LdStageArgs(ctypes.Structure):
    """vgpa_ld_stage_args (include/vgpa_hip.h)."""
    _fields_ = [("D", c_int32), ("row0", c_int32), ("Mp", c_int32), ("cw", c_int32), ("fwd", c_int32),
                ("kstore", c_int32), ("final_mode", c_int32), ("lda", c_int32), ("cx", c_double), ("cf", c_double),
                ("W", c_void_p), ("Wcol", c_void_p), ("E0", c_void_p), ("E1", c_void_p), ("J", c_void_p),
                ("base", c_void_p), ("K1", c_void_p), ("K23", c_void_p), ("out", c_void_p), ("A0", c_void_p),
                ("A1", c_void_p), ("x", c_void_p), ("e0", c_void_p), ("e1", c_void_p), ("jv", c_void_p),
                ("vbase", c_void_p), ("k1v", c_void_p), ("k23v", c_void_p), ("vout", c_void_p)]


COMM_COLLECTIVE = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p)
COMM_GROUP = ctypes.CFUNCTYPE(c_int, c_void_p)
COMM_P2P = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_uint64, c_int, c_void_p)
SHARD_OPT_GATHER_CHUNKS, SHARD_OPT_TIMEOUT_MS = 1, 2


class VgpaComm(ctypes.Structure):
    """vgpa_comm (include/vgpa_hip.h): the collectives of the row-sharded recursion as a table of function pointers."""
    _fields_ = [("user", c_void_p), ("all_gather", COMM_COLLECTIVE), ("all_to_all", COMM_COLLECTIVE),
                ("group_begin", COMM_GROUP), ("group_end", COMM_GROUP), ("send", COMM_P2P), ("recv", COMM_P2P),
                ("abort", COMM_GROUP)]


class VgpaShardProblem(ctypes.Structure):
    """vgpa_shard_problem (include/vgpa_hip.h): what the fused row-sharded sweep needs besides x (device pointers, obs_t host)."""
    _fields_ = [("theta", c_double), ("inv_sigma_diag", c_void_p), ("m0", c_void_p), ("s0", c_void_p), ("sigma", c_void_p),
                ("n_obs", c_int32), ("obs_t", POINTER(c_int64)), ("obs_y", c_void_p), ("obs_rinv_diag", c_void_p),
                ("obs_const", c_double), ("e0", c_double)]


class VgpaConfig(ctypes.Structure):
    _fields_ = [("abi_version", c_int32), ("device", c_int32), ("model", c_int32), ("method", c_int32),
                ("dim_d", c_int32), ("n_pts", c_int32), ("batch", c_int32), ("flags", c_int32),
                ("dt", c_double), ("n_theta", c_int32), ("n_obs", c_int32),
                ("theta", P_DOUBLE), ("sigma", P_DOUBLE), ("m0", P_DOUBLE), ("s0", P_DOUBLE),
                ("obs_t", POINTER(c_int64)), ("obs_y", P_DOUBLE), ("obs_noise", P_DOUBLE), ("obs_h", P_DOUBLE),
                ("e0", c_double)]


_lib = None


def load():
    """Loads (once) and returns the ctypes handle of libvgpa_hip.so; fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m vgpa_amd.build` "
                           "(vgpa_amd has no CPU fallback).")
    # PyTorch bundles its own HIP runtime with the same SONAME (libamdhip64.so.7).  When torch is used in
    # this process (bench.py, torch.distributed) it must be imported first so that one runtime is shared.
    if os.environ.get("VGPA_PRELOAD_TORCH", "0") == "1":
        import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    lib.vgpa_last_error.restype = c_char_p
    lib.vgpa_last_error.argtypes = [c_void_p]
    lib.vgpa_stream.restype = c_void_p
    lib.vgpa_stream.argtypes = [c_void_p]
    lib.vgpa_create.argtypes = [POINTER(c_void_p), POINTER(VgpaConfig)]
    lib.vgpa_destroy.argtypes = [c_void_p]
    lib.vgpa_destroy.restype = None
    lib.vgpa_synchronize.argtypes = [c_void_p]
    lib.vgpa_solve_fwd.argtypes = [c_void_p] + [c_void_p] * 7
    lib.vgpa_solve_bwd.argtypes = [c_void_p] + [c_void_p] * 7
    lib.vgpa_energy.argtypes = [c_void_p] + [c_void_p] * 9
    lib.vgpa_obs_energy.argtypes = [c_void_p] + [c_void_p] * 5
    lib.vgpa_energy_full.argtypes = [c_void_p] + [c_void_p] * 11
    lib.vgpa_free_energy.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.vgpa_gradient.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.vgpa_sweep.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vgpa_energy_parts.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vgpa_fetch.argtypes = [c_void_p, c_int, c_void_p]
    lib.vgpa_sweep_dev.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vgpa_free_energy_dev.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.vgpa_sweep_enqueue.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.vgpa_fetch_f.argtypes = [c_void_p, c_void_p]
    lib.vgpa_dev_alloc.argtypes = [c_void_p, c_uint64, POINTER(c_void_p)]
    lib.vgpa_dev_free.argtypes = [c_void_p, c_void_p]
    lib.vgpa_memcpy_h2d.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64]
    lib.vgpa_memcpy_d2h.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64]
    lib.vgpa_ld_gemm.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p, c_int]
    lib.vgpa_ld_stage.argtypes = [c_void_p, POINTER(LdStageArgs)]
    lib.vgpa_gradient_dev.argtypes = [c_void_p, c_void_p]
    lib.vgpa_release_x.argtypes = [c_void_p]
    lib.vgpa_shard_create.argtypes = [POINTER(c_void_p), c_int, c_double, c_int, c_int, c_int, c_int, c_int,
                                      POINTER(VgpaComm), c_void_p]
    lib.vgpa_shard_destroy.argtypes = [c_void_p]
    lib.vgpa_shard_destroy.restype = None
    lib.vgpa_shard_time_slice.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.vgpa_shard_stream.argtypes = [c_void_p]
    lib.vgpa_shard_stream.restype = c_void_p
    lib.vgpa_shard_synchronize.argtypes = [c_void_p]
    lib.vgpa_shard_solve_fwd.argtypes = [c_void_p] + [c_void_p] * 7
    lib.vgpa_shard_solve_bwd.argtypes = [c_void_p] + [c_void_p] * 7
    lib.vgpa_shard_sweep.argtypes = [c_void_p, POINTER(VgpaShardProblem), c_void_p, P_DOUBLE, c_void_p, c_void_p]
    lib.vgpa_shard_sweep.restype = c_int
    lib.vgpa_shard_sweep_sharded.argtypes = [c_void_p, POINTER(VgpaShardProblem), c_void_p, c_void_p, P_DOUBLE, c_void_p, c_void_p]
    lib.vgpa_shard_set_option.argtypes = [c_void_p, c_int, c_int64]
    lib.vgpa_shard_get_option.argtypes = [c_void_p, c_int, POINTER(c_int64)]
    lib.vgpa_shard_time_collectives.argtypes = [c_void_p, c_int, P_DOUBLE, P_DOUBLE]
    lib.vgpa_ld_gemm_chunk.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                       c_int, c_int, c_int]
    lib.vgpa_rccl_unique_id.argtypes = [c_void_p]
    lib.vgpa_rccl_comm_create.argtypes = [POINTER(VgpaComm), c_void_p, c_int, c_int, c_int]
    lib.vgpa_rccl_comm_destroy.argtypes = [POINTER(VgpaComm)]
    lib.vgpa_rccl_comm_destroy.restype = None
    lib.vgpa_rccl_comm_count.argtypes = [POINTER(VgpaComm), POINTER(c_int)]
    lib.vgpa_rccl_comm_streams.argtypes = [POINTER(VgpaComm), POINTER(c_int)]
    lib.vgpa_shard_time_stage.argtypes = [c_void_p, c_int, c_int, P_DOUBLE]
    lib.vgpa_shard_phase_ms.argtypes = [c_void_p, P_DOUBLE]
    lib.vgpa_time_slice.argtypes = [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]
    lib.vgpa_device_alloc.argtypes = [c_int, c_uint64, POINTER(c_void_p)]
    lib.vgpa_device_free.argtypes = [c_int, c_void_p]
    lib.vgpa_device_memcpy.argtypes = [c_int, c_void_p, c_void_p, c_uint64, c_int]
    lib.vgpa_vec_dot.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]
    lib.vgpa_vec_absmax.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p]
    lib.vgpa_vec_asum.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p]
    lib.vgpa_vec_axpby.argtypes = [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vgpa_set_option.argtypes = [c_void_p, c_int, c_int64]
    lib.vgpa_is_streaming.argtypes = [c_void_p]
    lib.vgpa_set_prior_energy.argtypes = [c_void_p, c_double]
    lib.vgpa_profile_begin.argtypes = [c_void_p]
    lib.vgpa_profile_end.argtypes = [c_void_p, P_DOUBLE, P_DOUBLE, P_DOUBLE, P_DOUBLE, POINTER(c_int64)]
    abi = lib.vgpa_abi_version()
    if (abi & 0xFFFF) != ABI_VERSION:
        raise RuntimeError(f"libvgpa_hip ABI {abi & 0xFFFF} != binding ABI {ABI_VERSION}: rebuild")
    if (abi & ABI_DIAGNOSTIC_BUILD) and os.environ.get("VGPA_ALLOW_DIAGNOSTIC") != "1":
        raise RuntimeError(f"{LIB_PATH} is a DIAGNOSTIC build (-DVGPA_EXPERIMENTS: wrong-result ablation switches and rejected "
                           "experiments compiled in); set VGPA_ALLOW_DIAGNOSTIC=1 to load it anyway, or rebuild with `python -m vgpa_amd.build`")
    _lib = lib
    return lib


def device_count():
    return int(load().vgpa_device_count())


def _raise(code, msg):
    msg = msg.decode() if isinstance(msg, bytes) else str(msg)
    if code == -1:
        raise ValueError(msg)
    if code == -3:
        raise np.linalg.LinAlgError(msg)
    if code == -5:
        raise NotImplementedError(msg)
    if code == -6:
        raise RuntimeError(f"libvgpa_hip: a collective of the row-sharded path failed or timed out; the communicator was aborted "
                           f"and this shard is unusable ({msg})")
    raise RuntimeError(f"libvgpa_hip error {code}: {msg}")


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class DeviceBuffer:
    """A caller-owned device allocation (fp64) on the context's device."""

    def __init__(self, ctx, count):
        self.ctx, self.count = ctx, int(count)
        p = c_void_p()
        ctx._check(ctx._lib.vgpa_dev_alloc(ctx._h, self.count * 8, byref(p)))
        self.ptr = p

    def upload(self, host):
        host = _c64(host).ravel()
        assert host.size == self.count
        self.ctx._check(self.ctx._lib.vgpa_memcpy_h2d(self.ctx._h, self.ptr, _ptr(host), host.size * 8))

    def upload_at(self, offset, host):
        """host -> this buffer from double `offset` on (large batches go up in pieces instead of through one host array)."""
        host = _c64(host).ravel()
        assert 0 <= offset and offset + host.size <= self.count
        dst = c_void_p(self.ptr.value + 8 * int(offset))
        self.ctx._check(self.ctx._lib.vgpa_memcpy_h2d(self.ctx._h, dst, _ptr(host), host.size * 8))

    def download_at(self, offset, count):
        out = np.empty(int(count))
        assert 0 <= offset and offset + out.size <= self.count
        src = c_void_p(self.ptr.value + 8 * int(offset))
        self.ctx._check(self.ctx._lib.vgpa_memcpy_d2h(self.ctx._h, _ptr(out), src, out.size * 8))
        return out

    def download(self):
        out = np.empty(self.count)
        self.ctx._check(self.ctx._lib.vgpa_memcpy_d2h(self.ctx._h, _ptr(out), self.ptr, out.size * 8))
        return out

    def free(self):
        if self.ptr and self.ctx._h is not None:
            self.ctx._lib.vgpa_dev_free(self.ctx._h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostArray:
    """What `DeviceArray.cpu()` returns: the data on the host (`.numpy()`), so that callers written against a tensor
    library's `.cpu().numpy()` read the same."""

    def __init__(self, a):
        self._a = a

    def numpy(self):
        return self._a


class DeviceArray:
    """An fp64 array in device memory owned by this object (vgpa_device_alloc; freed with it), with just enough surface for the
    callers of the row-sharded driver: `.ptr` / `.data_ptr()`, `.shape`, `[:n]` views of the leading dimension, `.numpy()` /
    `.cpu().numpy()` (download), and `__cuda_array_interface__`, through which a tensor library can wrap the memory without a copy
    (`torch.as_tensor(arr, device="cuda")`)."""

    def __init__(self, shape, device=0, _base=None, _ptr=None):
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.device = int(device)
        self._lib = load()
        self._base = _base                       # a view keeps its owner alive
        if _base is None:
            p = c_void_p()
            rc = self._lib.vgpa_device_alloc(self.device, max(self.size, 1) * 8, byref(p))
            if rc != 0:
                _raise(rc, f"vgpa_device_alloc({self.size * 8} bytes) failed")
            self.ptr = p.value
        else:
            self.ptr = int(_ptr)

    @property
    def size(self):
        n = 1
        for v in self.shape:
            n *= v
        return n

    @classmethod
    def from_host(cls, a, device=0):
        a = _c64(a)
        out = cls(a.shape, device)
        rc = out._lib.vgpa_device_memcpy(out.device, c_void_p(out.ptr), _ptr(a), a.size * 8, 1)
        if rc != 0:
            _raise(rc, "vgpa_device_memcpy (host -> device) failed")
        return out

    def data_ptr(self):
        return self.ptr

    def numpy(self):
        out = np.empty(self.shape)
        rc = self._lib.vgpa_device_memcpy(self.device, _ptr(out), c_void_p(self.ptr), out.size * 8, 2)
        if rc != 0:
            _raise(rc, "vgpa_device_memcpy (device -> host) failed")
        return out

    def cpu(self):
        return HostArray(self.numpy())

    def __getitem__(self, key):
        if not (isinstance(key, slice) and key.step in (None, 1)):
            raise TypeError("DeviceArray supports contiguous slices of the leading dimension only")
        lo, hi, _ = key.indices(self.shape[0])
        hi = max(hi, lo)
        inner = self.size // self.shape[0] if self.shape[0] else 0
        return DeviceArray((hi - lo,) + self.shape[1:], self.device, _base=self if self._base is None else self._base,
                           _ptr=self.ptr + 8 * lo * inner)

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 3, "strides": None}

    def __del__(self):
        try:
            if self._base is None and getattr(self, "ptr", None):
                self._lib.vgpa_device_free(self.device, c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


class ExternalBuffer:
    """A device allocation owned by someone else (e.g. a torch tensor's storage), passed by raw pointer."""

    def __init__(self, ptr, count):
        self.ptr, self.count = c_void_p(int(ptr)), int(count)


class Context:
    """
    Owns one vgpa_ctx.  `model` in {"NONE","OU","DW","L63","L96"}, `method` in {"euler","heun","rk2","rk4"}.
    Leave m0/s0/obs_* as None for an ODE-only context (solve_fwd / solve_bwd / energy operators).
    """

    def __init__(self, model, method, dim_d, n_pts, dt, sigma, theta=None, m0=None, s0=None, obs_t=None,
                 obs_y=None, obs_noise=None, obs_h=None, e0=0.0, batch=1, device=0, flags=0):
        self._lib = load()
        self._h = None
        self.model, self.method = str(model).upper(), str(method).lower()
        if self.model not in MODEL_IDS:
            raise ValueError(f" Unknown stochastic model -> {model}")
        if self.method not in METHOD_IDS:
            raise ValueError(f" Integration method is unknown -> {method}.")
        self.D, self.Np, self.B = int(dim_d), int(n_pts), int(batch)
        self.len_x = self.Np * self.D * (self.D + 1)
        keep = []

        def arr(a, n=None, dtype=np.float64):
            if a is None:
                return None
            a = np.ascontiguousarray(np.asarray(a, dtype=dtype)).ravel()
            if n is not None and a.size != n:
                raise ValueError(f"bad array size {a.size}, expected {n}")
            keep.append(a)
            return a

        d, dd = self.D, self.D * self.D
        theta_a = arr(np.atleast_1d(theta)) if theta is not None else None
        obs_t_a = arr(obs_t, dtype=np.int64)
        n_obs = 0 if obs_t_a is None else obs_t_a.size
        cfg = VgpaConfig()
        cfg.abi_version, cfg.device = ABI_VERSION, int(device)
        cfg.model, cfg.method = MODEL_IDS[self.model], METHOD_IDS[self.method]
        cfg.dim_d, cfg.n_pts, cfg.batch, cfg.flags = d, self.Np, self.B, int(flags)
        cfg.dt = float(dt)
        cfg.n_theta = 0 if theta_a is None else theta_a.size
        cfg.n_obs = n_obs
        cfg.e0 = float(e0)

        def dp(a):
            return None if a is None else a.ctypes.data_as(P_DOUBLE)

        cfg.theta = dp(theta_a)
        cfg.sigma = dp(arr(sigma, dd))
        cfg.m0 = dp(arr(m0, d))
        cfg.s0 = dp(arr(s0, dd))
        cfg.obs_t = None if obs_t_a is None else obs_t_a.ctypes.data_as(POINTER(c_int64))
        cfg.obs_y = dp(arr(obs_y, n_obs * d if obs_y is not None else None))
        cfg.obs_noise = dp(arr(obs_noise, dd))
        cfg.obs_h = dp(arr(obs_h, dd))
        h = c_void_p()
        rc = self._lib.vgpa_create(byref(h), byref(cfg))
        if rc != 0:
            _raise(rc, self._lib.vgpa_last_error(None))
        self._h = h
        self.n_obs = n_obs

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            _raise(rc, self._lib.vgpa_last_error(self._h))

    def close(self):
        if self._h is not None:
            self._lib.vgpa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape_v(self):
        return (self.B, self.Np, self.D)

    def _shape_m(self):
        return (self.B, self.Np, self.D, self.D)

    def _squeeze(self, a):
        return a[0] if self.B == 1 else a

    # ------------------------------------------------------------------ operator level
    def solve_fwd(self, lin_a, off_b, m0, s0, sigma):
        a, b = _c64(lin_a), _c64(off_b)
        m0, s0, sigma = _c64(m0), _c64(s0), _c64(sigma)
        mt, st = np.empty(self._shape_v()), np.empty(self._shape_m())
        self._check(self._lib.vgpa_solve_fwd(self._h, _ptr(a), _ptr(b), _ptr(m0), _ptr(s0), _ptr(sigma),
                                             _ptr(mt), _ptr(st)))
        return self._squeeze(mt), self._squeeze(st)

    def solve_bwd(self, lin_a, de_dm, de_ds, jm, js):
        a, em, es, jm, js = (_c64(v) for v in (lin_a, de_dm, de_ds, jm, js))
        lam, psi = np.empty(self._shape_v()), np.empty(self._shape_m())
        self._check(self._lib.vgpa_solve_bwd(self._h, _ptr(a), _ptr(em), _ptr(es), _ptr(jm), _ptr(js),
                                             _ptr(lam), _ptr(psi)))
        return self._squeeze(lam), self._squeeze(psi)

    def energy(self, lin_a, off_b, mt, st, want_edf=True, want_hyper=False):
        """(Esde, Ef, Edf, dEsde_dm, dEsde_ds[, dEsde_dth, dEsde_dSig]) -- the last two with want_hyper."""
        a, b, m, s = (_c64(v) for v in (lin_a, off_b, mt, st))
        esde = np.empty(self.B)
        ef, dm = np.empty(self._shape_v()), np.empty(self._shape_v())
        edf = np.empty(self._shape_m()) if want_edf else None
        ds = np.empty(self._shape_m())
        dth = dsg = None
        if want_hyper:
            dth = np.empty(self.B) if self.D == 1 else np.empty((self.B, self.D))
            dsg = np.empty(self.B) if self.D == 1 else np.empty((self.B, self.D, self.D))
        self._check(self._lib.vgpa_energy_full(self._h, _ptr(a), _ptr(b), _ptr(m), _ptr(s), _ptr(esde), _ptr(ef),
                                               _ptr(edf), _ptr(dm), _ptr(ds), _ptr(dth), _ptr(dsg)))
        sq = self._squeeze
        out = ((float(esde[0]) if self.B == 1 else esde), sq(ef), (sq(edf) if want_edf else None), sq(dm), sq(ds))
        if want_hyper:
            one = self.B == 1
            out += ((float(dth[0]) if self.D == 1 else dth[0]) if one else dth,
                    (float(dsg[0]) if self.D == 1 else dsg[0]) if one else dsg)
        return out

    def obs_energy(self, mt, st, want_jumps=True):
        m, s = _c64(mt), _c64(st)
        eobs = np.empty(self.B)
        jm = np.empty(self._shape_v()) if want_jumps else None
        js = np.empty(self._shape_m()) if want_jumps else None
        self._check(self._lib.vgpa_obs_energy(self._h, _ptr(m), _ptr(s), _ptr(eobs), _ptr(jm), _ptr(js)))
        e = float(eobs[0]) if self.B == 1 else eobs
        if not want_jumps:
            return e
        return e, self._squeeze(jm), self._squeeze(js)

    # ------------------------------------------------------------------ fused objective
    def free_energy(self, x):
        x = _c64(x)
        if x.size != self.B * self.len_x:
            raise ValueError(f"x has {x.size} entries, expected {self.B * self.len_x}")
        f = np.empty(self.B)
        self._check(self._lib.vgpa_free_energy(self._h, _ptr(x), _ptr(f)))
        return float(f[0]) if self.B == 1 else f

    def gradient(self, x=None):
        g = np.empty(self.B * self.len_x)
        xx = None if x is None else _c64(x)
        self._check(self._lib.vgpa_gradient(self._h, _ptr(xx), _ptr(g)))
        return g if self.B == 1 else g.reshape(self.B, self.len_x)

    def sweep(self, x):
        x = _c64(x)
        if x.size != self.B * self.len_x:
            raise ValueError(f"x has {x.size} entries, expected {self.B * self.len_x}")
        f, g = np.empty(self.B), np.empty(self.B * self.len_x)
        self._check(self._lib.vgpa_sweep(self._h, _ptr(x), _ptr(f), _ptr(g)))
        if self.B == 1:
            return float(f[0]), g
        return f, g.reshape(self.B, self.len_x)

    def energy_parts(self):
        e0, es, eo = np.empty(self.B), np.empty(self.B), np.empty(self.B)
        self._check(self._lib.vgpa_energy_parts(self._h, _ptr(e0), _ptr(es), _ptr(eo)))
        if self.B == 1:
            return float(e0[0]), float(es[0]), float(eo[0])
        return e0, es, eo

    def fetch(self, key):
        which = FETCH_IDS[key]
        if key in ("st", "psit", "Edf", "dEsde_ds"):
            out = np.empty(self._shape_m())
        elif key == "Esde_t":
            out = np.empty((self.B, self.Np))
        else:
            out = np.empty(self._shape_v())
        self._check(self._lib.vgpa_fetch(self._h, which, _ptr(out)))
        return self._squeeze(out)

    # ------------------------------------------------------------------ device-pointer path
    def alloc(self, count):
        return DeviceBuffer(self, count)

    def sweep_dev(self, x_buf, g_buf):
        f = np.empty(self.B)
        self._check(self._lib.vgpa_sweep_dev(self._h, x_buf.ptr, _ptr(f), g_buf.ptr))
        return float(f[0]) if self.B == 1 else f

    def free_energy_dev(self, x_buf):
        f = np.empty(self.B)
        self._check(self._lib.vgpa_free_energy_dev(self._h, x_buf.ptr, _ptr(f)))
        return float(f[0]) if self.B == 1 else f

    def sweep_enqueue(self, x_buf, g_buf):
        self._check(self._lib.vgpa_sweep_enqueue(self._h, x_buf.ptr, g_buf.ptr))

    def fetch_f(self):
        f = np.empty(self.B)
        self._check(self._lib.vgpa_fetch_f(self._h, _ptr(f)))
        return float(f[0]) if self.B == 1 else f

    def gradient_dev(self, g_buf):
        self._check(self._lib.vgpa_gradient_dev(self._h, g_buf.ptr))

    def release_x(self):
        """Tells the context that the device x of the last `*_dev` evaluation is about to be freed or overwritten."""
        self._check(self._lib.vgpa_release_x(self._h))

    # ------------------------------------------------------------------ device vector algebra (SCG)
    # Every vector is a DeviceBuffer of B segments; scalars are (B,) arrays, one per problem of the batch.
    def _seglen(self, buf):
        if buf.count % self.B:
            raise ValueError(f" vector of {buf.count} doubles does not split into {self.B} problems")
        return buf.count // self.B

    def _vreduce(self, fn, a, *b):
        out = np.empty(self.B)
        self._check(fn(self._h, a.ptr, *[v.ptr for v in b], self._seglen(a), _ptr(out)))
        return out

    def vdot(self, a, b):
        return self._vreduce(self._lib.vgpa_vec_dot, a, b)

    def vabsmax(self, a):
        return self._vreduce(self._lib.vgpa_vec_absmax, a)

    def vasum(self, a):
        return self._vreduce(self._lib.vgpa_vec_asum, a)

    def vaxpby(self, alpha, x, beta, y, out):
        """out = alpha*x + beta*y per problem (alpha, beta: scalars or (B,)); y=None drops the second term."""
        al = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float64), (self.B,)))
        be = None if y is None else np.ascontiguousarray(np.broadcast_to(np.asarray(beta, dtype=np.float64), (self.B,)))
        self._check(self._lib.vgpa_vec_axpby(self._h, self._seglen(x), _ptr(al), x.ptr,
                                             None if y is None else _ptr(be), None if y is None else y.ptr, out.ptr))

    def set_option(self, option, value):
        self._check(self._lib.vgpa_set_option(self._h, int(option), int(value)))

    def set_prior_energy(self, e0):
        """E0 = KL(q0||p0) used by the following objective calls (the reference recomputes it per call, variational.py:185)."""
        self._check(self._lib.vgpa_set_prior_energy(self._h, float(e0)))

    @property
    def streaming(self):
        return bool(self._lib.vgpa_is_streaming(self._h))

    def synchronize(self):
        self._check(self._lib.vgpa_synchronize(self._h))

    def profile_begin(self):
        self._check(self._lib.vgpa_profile_begin(self._h))

    def profile_end(self):
        v = [c_double() for _ in range(4)]
        n = c_int64()
        self._check(self._lib.vgpa_profile_end(self._h, byref(v[0]), byref(v[1]), byref(v[2]), byref(v[3]), byref(n)))
        return dict(fwd_ms=v[0].value, energy_ms=v[1].value, bwd_ms=v[2].value, grad_ms=v[3].value, n_sweeps=n.value)
