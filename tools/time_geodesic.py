"""Slab-mode geodesics (direct block-tridiagonal factorisation): set-up time, memory, time per batch of 64 fields and per single
field, against SuperLU on the host, for meshes of growing size.   python tools/time_geodesic.py [nu nv] ..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from animsnapbases_amd import GeodesicDistanceComputation, HipEngine
from oracle import asb_oracle as orc
from test_gpu_geodesic_pcg import _torus

sizes = [(475, 100), (1000, 100), (2000, 100)]
if len(sys.argv) > 2:
    sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for nu, nv in sizes:
    V, T = _torus(nu, nv)
    eng = HipEngine(0)
    t0 = time.perf_counter()
    geo = GeodesicDistanceComputation(V, T, engine=eng, backend="slab")
    geo.prepare()
    eng.sync()
    t_setup = time.perf_counter() - t0
    src = np.linspace(0, V.shape[0] - 1, 64).astype(np.int64)
    t0 = time.perf_counter(); eng.geodesic_solve(src, 1e-13); eng.sync(); t_b = time.perf_counter() - t0
    t0 = time.perf_counter(); phi, _ = eng.geodesic_solve(src, 1e-13); eng.sync(); t_b = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.geodesic_solve(src[:1], 1e-13); eng.sync(); t_1 = time.perf_counter() - t0
    line = "%7d vertices: %3d slabs (largest %d), set-up %.2f s, 64 fields %.1f ms, one field %.1f ms" % (
        V.shape[0], geo.n_slabs, geo.largest_slab, t_setup, t_b * 1e3, t_1 * 1e3)
    if V.shape[0] <= 120000:
        t0 = time.perf_counter(); ref = orc.Geodesics(V, T); t_h = time.perf_counter() - t0
        t0 = time.perf_counter(); r0 = ref(int(src[5])); t_h1 = time.perf_counter() - t0
        line += " | SuperLU: set-up %.2f s, one field %.1f ms, rel. difference %.1e" % (
            t_h, t_h1 * 1e3, np.linalg.norm(phi[5] - r0) / np.linalg.norm(r0))
    print(line, flush=True)
    eng.close()
