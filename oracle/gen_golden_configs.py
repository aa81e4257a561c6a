"""TEST INFRASTRUCTURE ONLY.  Full-size fixtures for BASELINE.json configs 2 and 3: runs the UNMODIFIED
reference (``/root/reference``) in the build container on the real rest meshes (``data/bunny.obj``,
``data/armadillo.obj``) with the seeded synthetic frames of SURVEY.md 8(d) and commits COMPACT results
(``tests/golden/c2_*.npz``, ``c3_*.npz``, each well under 2 MB):

    cd /root/repo && python oracle/gen_golden_configs.py [c2g c2l c3]

What a fixture holds: the rest mesh (vertex / triangle arrays parsed from the .obj -- data, not code), the
generator parameters (the frames are rebuilt from the seed by ``oracle.asb_oracle.synth_snapshots``), and of
the reference's outputs: the selected-vertex sequence, ``measures_at_largeDeforVerts``, ``pre_scale_factor``,
the CSV text, the values of ``comps`` / post-processed ``comps`` at a seeded sample of vertices, seeded random
projections of ``comps`` (K x 32) and ``weigs`` (32 x K), and for SPLOCS the printed energy trace, the centre
sequence of every outer iteration and the same sample / projections of the refined C, W after every outer
iteration.  The reference never travels; only these arrays do.
"""
import contextlib
import io
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import asb_oracle as orc          # only for the seeded *input* generator
from oracle.gen_golden import _param
from oracle.ref_import import REF_ROOT, import_reference

OUT = os.path.join(ROOT, "tests", "golden")
NPROJ = 32
NSAMPLE = 48


def read_obj(name):
    """Vertex / triangle arrays of a reference rest mesh (``v x y z`` / ``f a b c`` lines, 1-based)."""
    V, T = [], []
    with open(os.path.join(REF_ROOT, "data", name), encoding="latin-1") as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                V.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                T.append([int(q.split("/")[0]) - 1 for q in p[1:4]])
    return np.array(V, dtype=np.float64), np.array(T, dtype=np.int32)


def probes(N, F, seed=777):
    """Seeded probe matrices / sample sets shared by the generator and the tests."""
    rng = np.random.default_rng(seed)
    G = rng.normal(size=(3 * N, NPROJ)) / np.sqrt(3 * N)
    H = rng.normal(size=(NPROJ, F)) / np.sqrt(F)
    sv = np.sort(rng.choice(N, size=NSAMPLE, replace=False))
    return G, H, sv


def compact(comps, G, sv):
    K = comps.shape[0]
    return comps.reshape(K, -1) @ G, comps[:, sv, :].copy(), np.sqrt((comps.reshape(K, -1) ** 2).sum(1))


def run(ref, tag, mesh, F, rank, kind, seed, over, noise=1e-4):
    rest, tris = read_obj(mesh)
    verts = orc.synth_snapshots(rest, F, rank=rank, noise=noise, seed=seed, kind=kind)
    head = dict(rest=rest, tris=tris, F=np.array(F), rank=np.array(rank), kind=np.array(kind), seed=np.array(seed),
                noise=np.array(noise), mesh=np.array(mesh), probe_seed=np.array(777))
    run_arrays(ref, tag, verts, tris, over, head)


def run_c4(ref, tag="c4_uniform_pca_global", F=2000, N=100000, K=128, seed=4):
    verts, tris = orc.synth_uniform_snapshots(F, N, seed)
    head = dict(F=np.array(F), N=np.array(N), seed=np.array(seed), kind=np.array("uniform"), probe_seed=np.array(777),
                frame0_head=verts[0, :4].copy(), frame_last_tail=verts[-1, -4:].copy())
    run_arrays(ref, tag, verts, tris, dict(vertPos_numComponents=K), head)


def run_c5(ref, tag="c5_constraints_pod_deim", F=4000, ep=50000, K=256, rank=320, decay=0.955, noise=1e-9, seed=5):
    """Config 5: constraintsComponents.compute_pod_for_vectorized_nonlinear_snapshots_tensor
    (:298-320), post_process_components (:415-446; standardised, not orthogonalised) and deim (:797-860) of the UNMODIFIED
    reference on 50 000 x 3 rows x 4 000 frames.  The spectrum decays geometrically through all K components (rank 320,
    4.5 % per component, noise far below sigma_K), so every one of the K singular vectors is individually determined and
    the whole Pt sequence is comparable, not only the head."""
    import types
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    t0 = time.time()
    frames = orc.synth_constraint_frames(F, ep, rank, decay, noise, seed)
    print(tag, "input %.0f s" % (time.time() - t0), flush=True)
    M = 3 * ep
    rng = np.random.default_rng(778)
    Gp = rng.normal(size=(M, NPROJ)) / np.sqrt(M)
    sv = np.sort(rng.choice(ep, size=NSAMPLE, replace=False))
    param = types.SimpleNamespace(constProj_standarize=True, constProj_massWeight=False, constProj_orthogonal=False,
                                  deim_desired_num_components=K, constProj_output_directory=os.getcwd())
    ns = object.__new__(nonlinearSnapshots)
    ns.param = param
    ns.rest_shape = "first"
    ns.dim = 3
    ns.frs = F
    ns.constraintsSize = 1
    ns.num_constained_elements = ep
    ns.snapTensor = frames                          # the reference's read() leaves exactly this array (F, ep, 3) f64
    out = dict(F=np.array(F), ep=np.array(ep), K=np.array(K), rank=np.array(rank), decay=np.array(decay),
               noise=np.array(noise), seed=np.array(seed), probe_seed=np.array(778),
               frame0_head=frames[0, :4].copy(), frame_last_tail=frames[-1, -4:].copy())
    del frames
    ns.mean = None
    ns.pre_scale_factor = 1
    ns.massL = ns.invMassL = None
    ns.standarize()
    out["pre_scale_factor"] = np.float64(ns.pre_scale_factor)
    out["mean_proj"] = ns.mean.reshape(-1) @ Gp
    cc = object.__new__(constraintsComponents)
    cc.param = param
    cc.nonlinearSnapshots = ns
    cc.numComp = 0
    cc.comps = None
    cc.geom_interpol_verts = []
    rows = []
    writer = types.SimpleNamespace(writerow=lambda r: rows.append(list(r)))
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        cc.compute_pod_for_vectorized_nonlinear_snapshots_tensor(writer)
    out["ref_seconds"] = np.array(time.time() - t0)
    print(tag, "pod_vectorized %.0f s" % (time.time() - t0), flush=True)
    out["S"] = np.array([r[1] for r in rows])
    c = cc.comps.reshape(K, -1)
    out.update(comps_proj=c @ Gp, comps_sample=cc.comps[:, sv, :].copy(), sample_rows=sv,
               comps_gram_offdiag=np.abs(c @ c.T - np.eye(K)).max())
    with contextlib.redirect_stdout(io.StringIO()):
        cc.post_process_components()
    c = cc.comps.reshape(K, -1)
    out.update(post_proj=c @ Gp, post_sample=cc.comps[:, sv, :].copy(), post_norms=np.sqrt((c ** 2).sum(1)))
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        cc.deim()
    print(tag, "deim %.0f s" % (time.time() - t0), flush=True)
    out["Pt"] = np.asarray(cc.geom_Pt, dtype=np.int64)
    out["alpha"] = np.asarray(cc.geom_alpha, dtype=np.int64)
    out["alpha_ranges"] = np.asarray(cc.geom_alpha_ranges, dtype=np.int64)
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f KB" % (os.path.getsize(path) / 1024), "Pt[:8]", out["Pt"][:8].tolist(), flush=True)


def run_arrays(ref, tag, verts, tris, over, head):
    import snapbases.posComponents as pc_mod
    posSnapshots, posComponents = ref["posSnapshots"], ref["posComponents"]
    F, N = verts.shape[:2]
    G, H, sv = probes(N, F)
    work = os.getcwd()
    param = _param(vertPos_output_directory=work, name=tag, **over)
    K = param.vertPos_numComponents

    snap = object.__new__(posSnapshots)
    snap.input_animation_file = snap.input_test_animation_file = None
    snap.rest_shape = param.vertPos_rest_shape
    snap.verts = verts
    snap.test_verts = None
    snap.tris = tris.astype(np.int64)
    snap.test_tris = None
    snap.frs, snap.nVerts = F, N
    snap.mean = None
    snap.pre_scale_factor = 1
    snap.mass = snap.massL = snap.invMassL = None
    snap.snapTensor = None
    snap.compute_geodesic_distance = None
    snap.tet_mesh = None
    snap.massesFile = os.path.join(work, "none.bin")
    snap.read = lambda: None                       # the frames are already in memory (no h5py here)
    t0 = time.time()
    snap.do_snapshots_precomputations(param.q_standarize, param.q_massWeight)
    print(tag, "prepare %.1f s" % (time.time() - t0), flush=True)

    comp = object.__new__(posComponents)
    comp.basesType = param.vertPos_bases_type
    comp.pos_snapshots = snap
    comp.numComp = K
    comp.support = param.q_support
    comp.storeSingVal = True
    comp.comps = comp.weigs = comp.ortho_comps = None
    comp.smooth_min_dist = param.vertPos_smooth_min_dist
    comp.smooth_max_dist = param.vertPos_smooth_max_dist
    comp.output_components_file = "components.h5"
    comp.measures_at_largeDeforVerts = None
    comp.fileNameBases = "q_pos_"
    comp.param = param

    out = dict(head, pre_scale_factor=np.float64(snap.pre_scale_factor),
               mean_proj=snap.mean.reshape(-1) @ G, snap_proj=np.einsum("pf,fn->pn", H, snap.snapTensor.reshape(F, -1)) @ G)

    picked, geo_idx = [], []
    real_argmax = pc_mod.argmax

    def rec_argmax(a, *args, **kw):
        r = real_argmax(a, *args, **kw)
        picked.append(int(r))
        return r

    real_geo = snap.compute_geodesic_distance

    def rec_geo(idx):
        geo_idx.append(int(idx))
        return real_geo(idx)

    snap.compute_geodesic_distance = rec_geo
    splocs = param.vertPos_bases_type == "SPLOCS"
    last_prox, w_cols = [None], []
    it_C, it_W, it_G = [], [], []
    real_cho = pc_mod.cho_factor
    if splocs:
        real_prox, real_pw = posComponents.prox_l1l2, posComponents.project_weight
        admm = param.splocs_admm_num_itrs
        calls = [0]

        def rec_prox(Lambda, x, beta):
            z = real_prox(Lambda, x, beta)
            calls[0] += 1
            if calls[0] % admm == 0:               # the C an outer iteration ends with (posComponents.py:180)
                it_C.append(compact(z, G, sv))
                # weight columns of this outer iteration (one project_weight call per component unless one was skipped)
                it_W.append(H @ np.array(w_cols).T if len(w_cols) == K else np.full((NPROJ, K), np.nan))
                last_prox[0] = (Lambda[:, sv].copy(), Lambda.sum(1))
                del w_cols[:]
                print(tag, "outer iteration", len(it_C), "%.0f s" % (time.time() - t0), flush=True)
            return z

        def rec_pw(x):
            w = real_pw(x)
            w_cols.append(np.array(w, copy=True))
            return w

        def rec_cho(a, *args, **kw):               # a = W^T W + rho I (posComponents.py:170-172)
            it_G.append(np.array(a, copy=True) - param.splocs_rho * np.eye(a.shape[0]))
            if len(it_G) == 1:
                del w_cols[:2 * K]                 # drop the deflation loop's +-wk calls (local support: 2 per component)
            return real_cho(a, *args, **kw)

        comp.prox_l1l2 = rec_prox
        comp.project_weight = rec_pw
        pc_mod.cho_factor = rec_cho
    pc_mod.argmax = rec_argmax
    buf = io.StringIO()
    t0 = time.time()
    try:
        with contextlib.redirect_stdout(buf):
            comp.compute_components_store_singvalues()
    finally:
        pc_mod.argmax = real_argmax
        if splocs:
            pc_mod.cho_factor = real_cho
    print(tag, "compute_components_store_singvalues %.1f s" % (time.time() - t0), flush=True)
    out["ref_seconds"] = np.array(time.time() - t0)
    out["idx"] = np.array(picked[:K], dtype=np.int64)
    out["measures"] = comp.measures_at_largeDeforVerts.copy()
    P, S, nrm = compact(comp.comps, G, sv)
    out.update(comps_proj=P, comps_sample=S, comps_norms=nrm, sample_verts=sv,
               weigs_proj=H @ comp.weigs, weigs_norms=np.sqrt((comp.weigs ** 2).sum(0)),
               weigs_head=comp.weigs[:8].copy())
    out["csv_text"] = np.array(open(os.path.join(work, tag + "_posBases_pcaExtraction_singValues_errorNorm.csv")).read())
    if param.q_support == "local" or splocs:
        out["geo_idx"] = np.array(geo_idx, dtype=np.int64)
    if splocs:
        trace = []
        for line in buf.getvalue().splitlines():
            if line.startswith("itr "):
                parts = line.replace(",", " ").replace("=", " ").split()
                trace.append([float(parts[3]), float(parts[5])])
        out["splocs_trace"] = np.array(trace)
        out["splocs_centres"] = out["geo_idx"][K:].reshape(param.splocs_max_itrs, K)
        out["splocs_C_proj"] = np.array([c[0] for c in it_C])
        out["splocs_C_sample"] = np.array([c[1] for c in it_C])
        out["splocs_C_norms"] = np.array([c[2] for c in it_C])
        out["splocs_W_proj"] = np.array(it_W)
        out["splocs_WtW_last"] = it_G[-1]
        out["splocs_WtW_diag"] = np.array([np.diag(g) for g in it_G])
        out["splocs_Lambda_sample"], out["splocs_Lambda_rowsum"] = last_prox[0]
    with contextlib.redirect_stdout(io.StringIO()):
        comp.post_process_components()
    P, S, nrm = compact(comp.comps, G, sv)
    out.update(post_proj=P, post_sample=S, post_norms=nrm)
    for k, v in vars(param).items():
        if k != "vertPos_output_directory":
            out["param_" + k] = np.array(v)
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f KB" % (os.path.getsize(path) / 1024), "idx[:8]", out["idx"][:8].tolist(), flush=True)


CASES = {
    # config 2: bunny.obj, 200 frames, PCA K = 32 (global and local support)
    "c2g": ("c2_bunny_pca_global", "bunny.obj", 200, 20, "iid", 2, dict(vertPos_numComponents=32)),
    "c2l": ("c2_bunny_pca_local", "bunny.obj", 200, 20, "bumps", 2,
            dict(vertPos_numComponents=32, q_support="local", vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25)),
    # config 3: armadillo.obj, 1000 frames, PCA (local) + SPLOCS 20 x 10, lambda 2, rho 10, K = 64
    "c3": ("c3_armadillo_splocs", "armadillo.obj", 1000, 50, "bumps", 3,
           dict(vertPos_numComponents=64, q_support="local", vertPos_bases_type="SPLOCS",
                vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25,
                splocs_max_itrs=20, splocs_admm_num_itrs=10, splocs_lambda=2.0, splocs_rho=10.0)),
}


def main():
    which = sys.argv[1:] or list(CASES)
    ref = import_reference()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as work:
        os.chdir(work)                             # log_time writes function_timings.txt into cwd
        try:
            for w in which:
                if w == "c4":                      # ~20 min, ~25 GB: not part of the default list
                    run_c4(ref)
                    continue
                if w == "c5":                      # ~30 min, ~25 GB
                    run_c5(ref)
                    continue
                tag, mesh, F, rank, kind, seed, over = CASES[w]
                run(ref, tag, mesh, F, rank, kind, seed, over)
        finally:
            os.chdir(cwd)


if __name__ == "__main__":
    main()
