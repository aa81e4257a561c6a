import sys, time, types, contextlib, io
sys.path.insert(0, '.')
import numpy as np
from animsnapbases_amd import HipEngine
ep, F, K = 50000, 4000, 256
rng = np.random.default_rng(5)
frames = rng.normal(size=(F, 40)) @ rng.normal(size=(40, ep * 3))
frames = frames.reshape(F, ep, 3) + 1e-5 * rng.normal(size=(F, ep, 3))
e = HipEngine(0)
e.upload(frames, 0, ep)
e.sync()
for rep in range(2):
    t0 = time.perf_counter(); G = e.pod_gram(); t1 = time.perf_counter()
    lam, V = np.linalg.eigh(0.5 * (G + G.T)); t2 = time.perf_counter()
    S = np.sqrt(np.maximum(lam[::-1], 0)); V = V[:, ::-1]
    e.pod_basis(np.ascontiguousarray(V[:, :K]), S[:K]); e.sync(); t3 = time.perf_counter()
    print("gram (incl 128 MB D2H) %.3f s | host eigh %.3f s | back-projection %.3f s" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
