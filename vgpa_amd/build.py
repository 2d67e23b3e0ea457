"""
Builds vgpa_amd/lib/libvgpa_hip.so (the C-ABI HIP library) for gfx950 with hipcc.

    python -m vgpa_amd.build            # incremental
    python -m vgpa_amd.build --force

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libvgpa_hip.so")
SOURCES = ["vgpa_api.hip", "ode_generic.hip", "ode_mfma.hip", "ode_mfma_m0.hip", "ode_mfma_m1.hip", "ode_mfma_m2.hip",
           "ode_mfma_m3.hip", "ode_small.hip", "ode_wave.hip", "energy.hip", "assemble.hip", "large_d.hip", "large_d_stage.hip", "large_d_energy.hip", "vecops.hip", "host_linalg.cpp", "lib_gemm.cpp", "sharded_rccl.cpp"]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-ffp-contract=off"]


# per-file flags.  -amdgpu-mfma-vgpr-form: fp64-MFMA accumulators in VGPRs throughout (no v_accvgpr copies around the k loops); measured per
# translation unit -- the one-kernel stages and the energy-phase kernels above D = 64 gain 4 %, the stage products and the D <= 64 kernels
# do not (large_d_stage.hip)
FILE_CFLAGS = {"large_d_stage.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "large_d_energy.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}

# experiments: extra compiler flags (e.g. -DVGPA_SYM_SPLIT_C_FWD=1) without editing the sources; never set by the package itself
CFLAGS += os.environ.get("VGPA_EXTRA_CFLAGS", "").split()


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src, *extra))


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    import glob
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [HIPCC, *CFLAGS, *FILE_CFLAGS.get(s, []), "-c", src, "-o", obj]
            if s.endswith(".cpp"):
                cmd.insert(1, "-x"); cmd.insert(2, "hip")
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for: {failed}")
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
