#!/usr/bin/env python3
"""Times the BASELINE.json configurations that fit one MI355X (synthetic inputs of SURVEY.md 8d) and prints a
table (used for DESIGN.md section 5).  Not a parity test (tests/ hold those) -- sanity asserts only.

    python tools/run_configs.py [c1 c3 c5 ...]
"""
import contextlib
import io
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from animsnapbases_amd import constraintsComponents, nonlinearSnapshots, posComponents, posSnapshots   # noqa: E402
from oracle import asb_oracle as orc          # only for the seeded synthetic INPUT generators   # noqa: E402


def pos_param(K, support, kind="PCA", **kw):
    d = dict(vertPos_bases_type=kind, vertPos_numComponents=K, q_support=support, store_vertPos_PCA_sing_val=False,
             vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, q_standarize=True, q_massWeight=False,
             q_orthogonal=False, vertPos_output_directory=".", name="cfg", splocs_max_itrs=20, splocs_admm_num_itrs=10,
             splocs_lambda=2.0, splocs_rho=10.0)
    d.update(kw)
    return types.SimpleNamespace(**d)


def timed(fn):
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        r = fn()
    return time.perf_counter() - t0, r, buf.getvalue()


def run_pos(tag, rings, segs, F, K, support, kind):
    rest, tris = orc.synth_mesh(rings, segs, seed=1)
    verts = orc.synth_snapshots(rest, F, rank=min(50, F // 4), seed=1, kind="bumps" if support == "local" else "iid")
    t_prep, snaps, _ = timed(lambda: posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False))
    comp = posComponents(pos_param(K, support, kind), snaps)
    timed(comp.compute_components_store_singvalues) if kind == "PCA" else None        # warm-up (PCA only: SPLOCS is long)
    if snaps.compute_geodesic_distance is not None:
        snaps.compute_geodesic_distance._cache.clear()        # the timed call solves its geodesics itself
    t, _, out = timed(comp.compute_components_store_singvalues)
    assert np.isfinite(comp.measures_at_largeDeforVerts).all()
    extra = ""
    if kind == "SPLOCS":
        tr = comp.splocs_trace
        extra = " | SPLOCS energy %.6g -> %.6g, E_rms %.3e" % (tr[0, 0], tr[-1, 0], tr[-1, 1])
    print("%-4s N=%6d F=%5d K=%4d %-6s %-6s prepare %.3f s | compute_components_store_singvalues %.3f s -> %.1f snapshots/s%s"
          % (tag, rest.shape[0], F, K, kind, support, t_prep, t, F / t, extra), flush=True)


def run_c5(ep, F, K):
    rng = np.random.default_rng(5)
    r = 40
    frames = 0.1 + np.tensordot(rng.normal(size=(F, r)) * (0.85 ** np.arange(r))[None], rng.normal(size=(r, ep, 3)), (1, 0))
    frames += 1e-5 * rng.normal(size=(F, ep, 3))
    param = types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=1,
                                  constProj_massWeight=False, constProj_standarize=True, constProj_orthogonal=False,
                                  constProj_basis_type="pod_vectorized", deim_desired_num_components=K,
                                  constProj_store_sing_val=False, constProj_output_directory=".", name="c5", constProj_name="v")
    ns = nonlinearSnapshots(param, frames=frames)
    ns.config()
    t_prep, _, _ = timed(ns.snapshots_prepare)
    cc = constraintsComponents(param, ns)
    cc.config()
    t_pod, _, _ = timed(cc.compute_components_store_singvalues)
    t_post, _, _ = timed(cc.post_process_components)
    t_deim, _, _ = timed(cc.deim)
    assert len(set(cc.geom_Pt.tolist())) == K
    print("c5   rows=%d x3 F=%d K=%d  prepare %.3f s | POD %.3f s | post %.3f s | DEIM %.3f s -> %.1f snapshots/s (POD+DEIM)"
          % (ep, F, K, t_prep, t_pod, t_post, t_deim, F / (t_pod + t_deim)), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c1l", "c3", "c5s"]
    if "c1" in which:
        run_pos("c1/2", 76, 188, 200, 32, "global", "PCA")        # bunny-sized: 14 290 vertices
    if "c1l" in which:
        run_pos("c1L", 76, 188, 200, 32, "local", "PCA")
    if "c3" in which:
        run_pos("c3", 87, 170, 1000, 64, "local", "SPLOCS")       # armadillo-sized: 14 792 vertices
    if "c5s" in which:
        run_c5(12500, 1000, 64)                                   # quarter-scale config 5
    if "c5" in which:
        run_c5(50000, 4000, 256)
