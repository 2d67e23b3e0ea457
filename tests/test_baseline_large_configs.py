"""
GPU suite (-m gpu): BASELINE configs[3] (Lorenz-96 D = 1024) and configs[4] (D = 4096, row blocks over 8 ranks) at
their real matrix sizes, on short grids, against the numpy oracle.

  * `vgpa_ld_gemm` -- the fp64-MFMA product of every RK stage -- against numpy's dgemm for every kernel variant its
    dispatcher can pick: block-row tiles BM = 32 / 64 / 128 (chosen from the grid size), the 16-byte-load kernel
    (`k_gemm_v`: full tiles, even leading dimensions, 16-byte aligned operands), the full-tile scalar-load kernel
    (odd leading dimension or an 8-byte aligned pointer) and the bounds-checked one (ragged sizes), each with
    A or A^T, with and without the mid-point operand 0.5 (A0 + A1), plain and column-chunk packed output.
  * the fused sweep (F and gradient) at D = 1024, resident and time-chunked, vs `oracle.sweep(..., faithful=False)`.
  * the row-sharded recursion at D = 4096 with 8 virtual ranks (row blocks of 512, the packed all-to-all layout) vs
    the unsharded oracle recursion (src/numerics/runge_kutta4.py:79-109,179-207).
"""
import ctypes

import numpy as np
import pytest

from conftest import rel_err
from oracle import vgpa_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _tile_rows(M, N):
    """The dispatcher's choice (vgpa_amd/csrc/large_d.hip::launch_gemm): 128-row tiles when they still give two
    workgroups per CU, else 64-row tiles under the same condition, else 32."""
    nbx = (N + 63) // 64
    if nbx * ((M + 127) // 128) >= 512:
        return 128
    if nbx * ((M + 63) // 64) >= 512:
        return 64
    return 32


# (M, N, K, expected BM, what)
GEMM_SHAPES = [
    (1024, 1024, 1024, 32, "configs[3] stage product"),
    (1536, 1536, 1536, 64, "64-row tiles"),
    (2048, 2048, 2048, 128, "128-row tiles"),
    (2048, 4096, 4096, 128, "two ranks' rows of configs[4]"),
    (512, 4096, 4096, 64, "one of eight ranks' row block of configs[4]"),
]
RAGGED_SHAPES = [
    (1000, 1030, 1001, 32, "ragged, 32-row tiles"),
    (1540, 1530, 1525, 64, "ragged, 64-row tiles"),
    (2050, 2060, 2049, 128, "ragged, 128-row tiles"),
]


def _run_gemm(transa, M, N, K, mid, lda_pad, misalign, cw_chunks, seed):
    import torch
    from legacy_sharded import HipStageBackend
    be = HipStageBackend()
    rng = np.random.default_rng(seed)
    rows, cols = (K, M) if transa else (M, K)            # storage of A (A^T product: A is [K][M])
    lda, ldb = cols + lda_pad, N + lda_pad
    dev = torch.device("cuda", 0)

    def dev_matrix(r, ld, c):
        host = np.zeros((r, ld))
        host[:, :c] = rng.standard_normal((r, c))
        flat = torch.zeros(r * ld + 2, dtype=torch.float64, device=dev)
        off = 1 if misalign else 0                        # 8-byte aligned only: not eligible for 16-byte loads
        flat[off:off + r * ld] = torch.as_tensor(host.ravel(), device=dev)
        return host[:, :c], flat, off

    a0, a0_d, off = dev_matrix(rows, lda, cols)
    a1 = a1_d = None
    if mid:
        a1, a1_d, _ = dev_matrix(rows, lda, cols)
    b, b_d, boff = dev_matrix(K, ldb, N)
    cw = N // cw_chunks
    c_d = torch.full((M * N,), float("nan"), dtype=torch.float64, device=dev)
    # vgpa_ld_gemm takes raw pointers: pass B through its own offset
    lib = be._lib
    rc = lib.vgpa_ld_gemm(be._stream(), int(transa), M, N, K, be._p(a0_d, off), be._p(a1_d, off) if mid else None, lda,
                          be._p(b_d, boff), ldb, be._p(c_d), cw)
    assert rc == 0
    torch.cuda.synchronize()
    a = a0 if not mid else 0.5 * (a0 + a1)
    want = (a.T if transa else a).dot(b)
    got = c_d.cpu().numpy().reshape(cw_chunks, M, cw)
    got = np.concatenate([got[q] for q in range(cw_chunks)], axis=1)
    return got, want


@pytest.mark.parametrize("transa", [False, True])
@pytest.mark.parametrize("M,N,K,bm,what", GEMM_SHAPES)
def test_stage_gemm_full_tiles(M, N, K, bm, what, transa):
    """Full tiles: the 16-byte-load kernel (aligned, even ld), and the scalar-load full-tile kernel reached through an
    odd leading dimension and through an 8-byte aligned pointer; mid-point operand; packed output."""
    assert _tile_rows(M, N) == bm
    for mid, lda_pad, misalign, chunks in ((False, 0, False, 1), (True, 0, False, 1), (False, 1, False, 1),
                                           (False, 0, True, 1), (True, 0, False, N // 512)):
        got, want = _run_gemm(transa, M, N, K, mid, lda_pad, misalign, chunks, seed=M + N + K + mid)
        assert rel_err(got, want) < 1e-12, (what, mid, lda_pad, misalign, chunks)


@pytest.mark.parametrize("transa", [False, True])
@pytest.mark.parametrize("M,N,K,bm,what", RAGGED_SHAPES)
def test_stage_gemm_ragged(M, N, K, bm, what, transa):
    assert _tile_rows(M, N) == bm
    for mid in (False, True):
        got, want = _run_gemm(transa, M, N, K, mid, 3, False, 1, seed=M + K + mid)
        assert rel_err(got, want) < 1e-12, (what, mid)


def test_config4_fused_sweep_at_d1024():
    """BASELINE configs[3]: Lorenz-96, D = 1024, RK4 -- F, the gradient and the state arrays of the fused sweep on a
    4-point grid, resident and time-chunked (chunk of one step), against the oracle."""
    from test_gpu_edge_cases import make_problem, gpu_context
    from vgpa_amd._lib import FLAG_STREAM_LARGE_D, OPT_LD_CHUNK
    p, x = make_problem("L96", 1024, 4, method="rk4", obs_at=[2])
    f_o, g_o, st_o = vo.sweep(p, x, faithful=False)
    res = gpu_context(p)
    assert not res.streaming
    f, g = res.sweep(x)
    assert abs(f - f_o) <= TOL * abs(f_o), (f, f_o)
    assert rel_err(g, g_o) < TOL
    for key in ("mt", "st", "lamt", "psit"):
        assert rel_err(res.fetch(key), st_o[key]) < TOL, key
    res.close()
    stc = gpu_context(p, flags=FLAG_STREAM_LARGE_D)
    assert stc.streaming
    stc.set_option(OPT_LD_CHUNK, 1)
    f_s, g_s = stc.sweep(x)
    assert f_s == f and np.array_equal(g_s, g)                  # same kernels, same order per grid point
    stc.close()


def test_config5_row_sharded_recursion_at_d4096_eight_ranks():
    """BASELINE configs[4]: D = 4096, RK4, rows sharded over 8 ranks (blocks of 512 rows; 64-row GEMM tiles, packed
    column-chunk output, all-to-all + grouped all-gather per stage) on the NATIVE driver (vgpa_shard_solve_fwd / _bwd: the
    step / stage loop and the collectives inside libvgpa_hip.so).  Eight virtual ranks share the one GPU of the test box;
    one forward and one backward RK4 step; every rank's time slice against the unsharded oracle."""
    from test_large_d import _run_native_virtual_ranks
    _run_native_virtual_ranks("rk4", 4096, 2, 8)


def test_config4_time_chunked_sweep_at_d1024_with_a_real_chunk():
    """BASELINE configs[3]'s matrix size on a grid long enough for the time-chunked sweep to mean something: D = 1024, RK4,
    Np = 33, chunks of 8 steps (four full chunks: Psi_t and dEsde_dS exist only as chunk ring buffers, the backward recursion
    crosses three chunk boundaries, eight observations fall inside and across chunks) -- against ORACLE anchors generated in the
    build container (tools/gen_d4096_anchor.py 1024 33 -> tests/golden/anchors_d1024_np33.json, ~1 minute of CPU): F, per-grid-
    point norms / maxima / sampled entries of the gradient, state norms.  Then the resident sweep of the same context size must
    equal the chunked one bit for bit, as must a chunk length that does not divide the grid (5)."""
    import json
    import os
    import sys
    from conftest import ROOT, GOLDEN_DIR
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_d4096_anchor import inputs_grid
    from vgpa_amd._lib import FLAG_STREAM_LARGE_D, OPT_LD_CHUNK
    import vgpa_amd as va
    a = json.load(open(os.path.join(GOLDEN_DIR, "anchors_d1024_np33.json")))
    d, n = a["D"], a["Np"]
    x, m0, s0, sig, obs_t, obs_y, rdiag = inputs_grid(d, n)

    def context(flags):
        return va.Context("L96", a["method"], d, n, a["dt"], sigma=np.diag(sig), theta=[8.0], m0=m0, s0=s0, obs_t=obs_t, obs_y=obs_y,
                          obs_noise=np.diag(rdiag), e0=0.0, flags=flags)

    ctx = context(FLAG_STREAM_LARGE_D)
    assert ctx.streaming
    ctx.set_option(OPT_LD_CHUNK, 8)
    f, g = ctx.sweep(x)
    ga, gb = g[:n * d * d].reshape(n, d, d), g[n * d * d:].reshape(n, d)
    assert abs(f - a["F_minus_E0"]) <= TOL * abs(a["F_minus_E0"])
    e0, esde, eobs = ctx.energy_parts()
    assert abs(esde - a["Esde"]) <= TOL * abs(a["Esde"]) and abs(eobs - a["Eobs"]) <= TOL * abs(a["Eobs"])
    for t in range(n):
        assert abs(np.linalg.norm(ga[t]) - a["grad_a_fro"][t]) <= TOL * a["grad_a_fro"][t], t
        assert abs(np.abs(ga[t]).max() - a["grad_a_absmax"][t]) <= TOL * a["grad_a_absmax"][t], t
        assert abs(np.linalg.norm(gb[t]) - a["grad_b_norm"][t]) <= TOL * a["grad_b_norm"][t], t
        got = np.array([ga[t][i, j] for (i, j) in a["samples_ij"]])
        assert rel_err(got, np.array(a["grad_a_samples"][t])) < 1e-8, t            # (single entries, some of them near cancellation)
        assert rel_err(gb[t][:8], np.array(a["grad_b_first8"][t])) < TOL, t
    st, mt, lam = ctx.fetch("st"), ctx.fetch("mt"), ctx.fetch("lamt")
    for t in range(n):
        assert abs(np.linalg.norm(st[t]) - a["st_fro"][t]) <= TOL * a["st_fro"][t]
        assert abs(np.linalg.norm(mt[t]) - a["mt_norm"][t]) <= TOL * a["mt_norm"][t]
        assert abs(np.linalg.norm(lam[t]) - a["lamt_norm"][t]) <= TOL * max(a["lamt_norm"][t], 1e-300)
    ctx.close()
    odd = context(FLAG_STREAM_LARGE_D)
    odd.set_option(OPT_LD_CHUNK, 5)
    f5, g5 = odd.sweep(x)
    odd.close()
    assert f5 == f and np.array_equal(g5, g)
    res = context(0)
    assert not res.streaming
    f_r, g_r = res.sweep(x)
    assert f_r == f and np.array_equal(g_r, g)                  # same kernels, same order per grid point
    psit = res.fetch("psit")
    for t in range(n):
        assert abs(np.linalg.norm(psit[t]) - a["psit_fro"][t]) <= TOL * max(a["psit_fro"][t], 1e-300)
    res.close()
