"""Residual loop vs panel algorithm on one rank for growing tensors (where is the crossover?)."""
import sys, time, contextlib, io, types
sys.path.insert(0, '.')
import numpy as np
from animsnapbases_amd import HipEngine
rng = np.random.default_rng(0)
N, K = 14290, 32
for F in (200, 400, 800, 1600):
    base = rng.normal(size=(N, 3))
    X = base[None] + np.tensordot(rng.normal(size=(F, 30)) * (0.8 ** np.arange(30)), rng.normal(size=(30, N, 3)) * 0.02, (1, 0)) + 1e-4 * rng.normal(size=(F, N, 3))
    X = (X - X[0:1])
    out = []
    for mode in (0, 1):
        e = HipEngine(0)
        e.upload(X, 0, N)
        ts = []
        for rep in range(3):
            e.deflate_begin(K, False, mode); e.sync()
            t = time.perf_counter(); e.run_global(0, K); e.sync(); ts.append(time.perf_counter() - t)
        e.close()
        out.append(min(ts) * 1e3)
    print("F=%4d  tensor %4.0f MB  residual %.2f ms  panels %.2f ms" % (F, 24 * N * ((F + 15) // 16 * 16) / 1e6, out[0], out[1]), flush=True)
