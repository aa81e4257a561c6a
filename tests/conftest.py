import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)


def align_signs(comps, weigs, ref_comps):
    """Global-support components are defined up to a joint sign of (comps[k], weigs[:,k])
    (LAPACK's SVD sign is arbitrary, SURVEY.md section 7 'SVD sign').  Flip to match."""
    comps, weigs = comps.copy(), weigs.copy()
    for k in range(comps.shape[0]):
        if np.vdot(comps[k], ref_comps[k]) < 0:
            comps[k] *= -1
            weigs[:, k] *= -1
    return comps, weigs


@pytest.fixture(scope="session")
def golden():
    return load_golden
