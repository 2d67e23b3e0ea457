"""Writes tests/golden/h5py_gzip_result.h5 with h5py exactly as the reference saves its results
(simulation.py:290-307: create_dataset(key, data=..., shape=..., compression="gzip")) plus the expected arrays.
Run with an interpreter that has h5py (here: /opt/conda/bin/python3.9 tools/gen_h5_fixture.py)."""
import h5py, numpy as np
rng = np.random.default_rng(7)
d = {"at": rng.standard_normal((21, 3, 3)), "bt": rng.standard_normal((21, 3)), "fx": np.atleast_1d(12.5),
     "m0": rng.standard_normal(3), "s0": 0.2 * np.eye(3), "mt": rng.standard_normal((21, 3)),
     "st": rng.standard_normal((21, 3, 3)), "lamt": rng.standard_normal((21, 3)), "psit": rng.standard_normal((21, 3, 3)),
     "Efx": rng.standard_normal((21, 3)), "Edf": rng.standard_normal((21, 3, 3))}
with h5py.File("/root/repo/tests/golden/h5py_gzip_result.h5", "w") as f:        # exactly the reference's save() calls
    for k in d:
        f.create_dataset(k, data=d[k], shape=d[k].shape, compression="gzip")
np.savez_compressed("/root/repo/tests/golden/h5py_gzip_result_expected.npz", **d)
print("fixture written")
