"""Config-5 POD step by step on the GPU box: Gram, eigen-problem (device tridiagonalisation vs host eigh), back-projection."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from scipy.linalg import eigh_tridiagonal
from animsnapbases_amd import HipEngine
ep, F, K = 50000, 4000, 256
if len(sys.argv) > 1:
    ep, F, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(5)
frames = rng.normal(size=(F, 40)) @ rng.normal(size=(40, ep * 3))
frames = frames.reshape(F, ep, 3) + 1e-5 * rng.normal(size=(F, ep, 3))
e = HipEngine(0)
e.upload(frames, 0, ep)
e.sync()
for rep in range(2):
    t0 = time.perf_counter(); e.pod_gram(to_host=False); e.sync(); t1 = time.perf_counter()
    d, off = e.sym_tridiag(F); t2 = time.perf_counter()
    lam = eigh_tridiagonal(d, off, eigvals_only=True)[::-1]; t3 = time.perf_counter()
    lk, Z = eigh_tridiagonal(d, off, select='i', select_range=(F - K, F - 1), lapack_driver='stemr'); t4 = time.perf_counter()
    V = e.sym_backtransform(F, Z[:, ::-1]); t5 = time.perf_counter()
    S = np.sqrt(np.maximum(lam, 0))
    e.pod_basis(np.ascontiguousarray(V), S[:K]); e.sync(); t6 = time.perf_counter()
    print("gram %.3f s | tridiag (device) %.3f s | eigvals(T) %.3f s | %d vectors of T %.3f s | back-transform %.3f s | "
          "back-projection %.3f s || total %.3f s" % (t1 - t0, t2 - t1, t3 - t2, K, t4 - t3, t5 - t4, t6 - t5, t6 - t0), flush=True)
t0 = time.perf_counter(); G = e.pod_gram(); t1 = time.perf_counter()
lamh, Vh = np.linalg.eigh(0.5 * (G + G.T)); t2 = time.perf_counter()
print("host path: gram + 128 MB D2H %.3f s | host eigh %.3f s" % (t1 - t0, t2 - t1))
print("eigenvalue agreement (top K, relative):", np.abs(lam[:K] - lamh[::-1][:K]).max() / lamh[-1])
Vh = Vh[:, ::-1][:, :K]
sgn = np.sign(np.sum(V * Vh, axis=0))
print("eigenvector agreement (top 40 = the data's rank):", np.abs(V[:, :40] * sgn[None, :40] - Vh[:, :40]).max())
