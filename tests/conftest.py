import os
import sys
import glob

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def golden_tags():
    """Sweep fixtures (one per model / stepper case); host_terms.npz holds the host-side helper vectors."""
    tags = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [t for t in tags if t not in ("host_terms", "h5py_gzip_result_expected")]


def load_golden(tag):
    with np.load(os.path.join(GOLDEN_DIR, f"{tag}.npz")) as z:
        return {k: z[k] for k in z.files}


def rel_err(got, want):
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    scale = max(float(np.max(np.abs(want))) if want.size else 0.0, 1e-300)
    return float(np.max(np.abs(got - want))) / scale if want.size else 0.0


@pytest.fixture(params=golden_tags())
def golden(request):
    z = load_golden(request.param)
    z["_tag"] = request.param
    return z
