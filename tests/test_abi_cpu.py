"""CPU: the C-ABI library loads, exports every symbol include/asb.h declares, and the
product fails loudly (no fallback) when no gfx950 GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from animsnapbases_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "asb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(asb_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    syms = _declared_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libasb_hip.so does not export %s" % s
    assert sorted(_lib.PROTOTYPES) == syms, "ctypes prototypes and include/asb.h disagree"
    assert _lib.load().asb_abi_version() == 1


def test_eig3_host_probe():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for _ in range(200):
        S = rng.normal(size=(3, 40)) * rng.uniform(1e-3, 10, size=(3, 1))
        G = S @ S.T
        a6 = np.array([G[0, 0], G[0, 1], G[0, 2], G[1, 1], G[1, 2], G[2, 2]])
        out = np.zeros(4)
        lib.asb_test_eig3(a6.ctypes.data, out.ctypes.data)
        w, V = np.linalg.eigh(G)
        u = V[:, -1] * np.sign(V[:, -1] @ out[1:])
        assert abs(out[0] - w[-1]) <= 1e-13 * w[-1]
        assert np.abs(u - out[1:]).max() < 1e-9
        assert out[1:][np.argmax(np.abs(out[1:]))] > 0        # canonical sign
    # degenerate: zero matrix must not produce NaN
    out = np.zeros(4)
    lib.asb_test_eig3(np.zeros(6).ctypes.data, out.ctypes.data)
    assert np.isfinite(out).all() and out[0] == 0.0


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from animsnapbases_amd import AsbLibraryError, HipEngine, posSnapshots
    with pytest.raises(AsbLibraryError):
        HipEngine(0)
    with pytest.raises(AsbLibraryError):
        posSnapshots.from_arrays(np.zeros((4, 5, 3)), None, "first")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "animsnapbases_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f
