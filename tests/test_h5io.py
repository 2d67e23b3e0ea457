"""Result files: the self-contained HDF5 writer / reader against h5py-written data."""
import os
import subprocess
import sys

import numpy as np
import pytest

import vgpa_amd as va
from conftest import GOLDEN_DIR

CONDA_PY = "/opt/conda/bin/python3.9"          # has h5py in the build image; tests that need it skip elsewhere


def _sample(rng):
    return {"at": rng.standard_normal((9, 5, 5)), "bt": rng.standard_normal((9, 5)), "fx": -3.5,
            "m0": rng.standard_normal(5), "s0": 0.2 * np.eye(5), "mt": rng.standard_normal((9, 5)),
            "st": rng.standard_normal((9, 5, 5)), "obs_t": np.arange(4), "mask": np.array([True, False]),
            "single": np.float32(1.5) * np.ones(3, dtype=np.float32), "empty": np.zeros((0, 3))}


def test_roundtrip(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "h5py", None)             # force the built-in reader even where h5py exists
    data = _sample(np.random.default_rng(1))
    path = tmp_path / "out.h5"
    va.save_h5(path, data)
    back = va.load_h5(path)
    assert set(back) == set(data)
    for key, val in data.items():
        want = np.atleast_1d(val)
        want = want.astype(np.uint8) if want.dtype == bool else want
        assert back[key].shape == want.shape and np.array_equal(back[key], want), key
        assert back[key].dtype == want.dtype, key
    assert back["fx"].shape == (1,)                            # scalars become shape-(1,) arrays (simulation.py:298-300)


def test_reader_parses_the_reference_style_file(monkeypatch):
    """tests/golden/h5py_gzip_result.h5 was written by h5py with the reference's create_dataset(..., 'gzip') calls."""
    monkeypatch.setitem(sys.modules, "h5py", None)
    got = va.load_results(os.path.join(GOLDEN_DIR, "h5py_gzip_result.h5"))
    with np.load(os.path.join(GOLDEN_DIR, "h5py_gzip_result_expected.npz")) as want:
        assert set(got) == set(want.files)
        for key in want.files:
            assert got[key].dtype == want[key].dtype and np.array_equal(got[key], want[key]), key


def test_rejects_what_it_cannot_read(tmp_path):
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"not an hdf5 file at all")
    with pytest.raises((ValueError, OSError)):
        va.load_h5(bad)
    with pytest.raises(TypeError):
        va.save_h5(tmp_path / "x.h5", {"s": np.array(["text"])})
    with pytest.raises(RuntimeError):
        va.load_results(None)


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with h5py on this machine")
def test_h5py_opens_the_writers_files(tmp_path):
    data = _sample(np.random.default_rng(2))
    path, ref = tmp_path / "mine.h5", tmp_path / "ref.npz"
    va.save_h5(path, data)
    np.savez(ref, **{k: np.atleast_1d(v).astype(np.uint8) if np.atleast_1d(v).dtype == bool else np.atleast_1d(v)
                     for k, v in data.items()})
    code = ("import h5py, numpy as np, sys\n"
            "z = np.load(sys.argv[2])\n"
            "with h5py.File(sys.argv[1], 'r') as f:\n"
            "    assert set(f.keys()) == set(z.files)\n"
            "    for k in z.files:\n"
            "        a = np.array(f[k])\n"
            "        assert a.dtype == z[k].dtype and a.shape == z[k].shape and np.array_equal(a, z[k]), k\n"
            "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    res = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, str(path), str(ref)], capture_output=True, text=True, env=env)
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr


@pytest.mark.gpu
def test_optimise_save_load(tmp_path, monkeypatch):
    """optimise -> save_results -> load_results: the reference's result-file key set (simulation.py:290-307)."""
    from helpers import build_problem
    monkeypatch.chdir(tmp_path)
    outs = []
    for resident in (False, True):
        p = build_problem("OU", "Euler", 2.0, 0.01, None)
        v = p["vgp"]
        opts = {"max_it": 8, "x_tol": 1e-6, "f_tol": 1e-8, "display": False}
        opt = v.device_scg(opts) if resident else va.SCG(v.free_energy, v.gradient, opts)
        x, fx = opt(v.initialization())
        path, written = va.save_results("OU run %d" % resident, v, x, fx)
        assert str(path) == "OU_run_%d.h5" % resident
        back = va.load_results(path)
        assert set(back) == {"at", "bt", "fx", "m0", "s0", "mt", "st", "lamt", "psit", "Efx", "Edf"}
        for key, val in written.items():
            assert np.array_equal(back[key], np.atleast_1d(val)), key
        outs.append(back)
    assert abs(outs[0]["fx"][0] - outs[1]["fx"][0]) <= 1e-9 * abs(outs[0]["fx"][0])
    assert np.allclose(outs[0]["mt"], outs[1]["mt"], rtol=1e-7, atol=1e-9)
    with pytest.raises(RuntimeError):
        va.load_results(None)
