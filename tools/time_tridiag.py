"""Tridiagonalisation of an n x n Gram-like matrix: ms per call, panel kernel against the two-launch loop (ASB_TD_PANEL_MIN=0 in a
second process), and the spectrum of T against numpy.   python tools/time_tridiag.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from scipy.linalg import eigh_tridiagonal
from animsnapbases_amd import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(1)
Bm = rng.normal(size=(n, 40)) * (0.85 ** np.arange(40))[None]
A = Bm @ Bm.T + 1e-6 * (lambda N: N @ N.T)(rng.normal(size=(n, n))) / n
A = 0.5 * (A + A.T)
e = HipEngine(0, stream=0)
ts = []
for rep in range(3):
    Ad = torch.from_numpy(A.copy()).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d, off = e.sym_tridiag(n, Ad.data_ptr())
    ts.append((time.perf_counter() - t0) * 1e3)
lam = eigh_tridiagonal(d, off, eigvals_only=True)
ref = np.linalg.eigvalsh(A)
print("n = %d, ASB_TD_PANEL_MIN=%s: %s ms; spectrum of T against numpy: %.2e of the largest eigenvalue" % (
    n, os.environ.get("ASB_TD_PANEL_MIN", "default"), ", ".join("%.1f" % t for t in ts), np.abs(lam - ref).max() / abs(ref).max()))
e.close()
