"""TEST INFRASTRUCTURE ONLY.  Generates ``tests/golden/*.npz`` by running the UNMODIFIED
reference (``/root/reference``) in the build container on small seeded inputs.

    cd /root/repo && python oracle/gen_golden.py

The reference itself never travels: only the inputs and the arrays it produced are
committed.  The reference classes are instantiated with ``object.__new__`` and the
attributes their ``__init__`` would set (this avoids ``.h5`` files / libigl, which the
image lacks); every *method* that runs is the reference's own.
Selected-vertex indices are observed by wrapping the ``argmax`` name inside the
reference module's namespace (a recording pass-through), SPLOCS internals by
instance-level pass-through wrappers around ``prox_l1l2`` / ``project_weight``.
"""
import contextlib
import hashlib
import io
import os
import struct
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import asb_oracle as orc          # only for the seeded *input* generators
from oracle.ref_import import import_reference

OUT = os.path.join(ROOT, "tests", "golden")


def _param(**kw):
    base = dict(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                q_support="global", vertPos_numComponents=4, store_vertPos_PCA_sing_val=True,
                vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25,
                splocs_max_itrs=3, splocs_admm_num_itrs=4, splocs_lambda=2.0, splocs_rho=10.0,
                vertPos_rest_shape="first", name="golden", vertPos_output_directory=".")
    base.update(kw)
    return types.SimpleNamespace(**base)


def run_reference_pos(ref, verts, tris, param, mass=None, workdir="."):
    """Drives reference posSnapshots + posComponents on in-memory data."""
    posSnapshots, posComponents = ref["posSnapshots"], ref["posComponents"]
    import snapbases.posComponents as pc_mod

    snap = object.__new__(posSnapshots)
    snap.input_animation_file = snap.input_test_animation_file = None
    snap.rest_shape = param.vertPos_rest_shape
    snap.verts = verts.astype(float)
    snap.test_verts = None
    snap.tris = tris
    snap.test_tris = None
    snap.frs, snap.nVerts = verts.shape[0], verts.shape[1]
    snap.mean = None
    snap.pre_scale_factor = 1
    snap.mass = snap.massL = snap.invMassL = None
    snap.snapTensor = None
    snap.compute_geodesic_distance = None
    snap.tet_mesh = None
    snap.massesFile = os.path.join(workdir, "mass.bin")
    if mass is not None:
        with open(snap.massesFile, "wb") as fh:
            fh.write(struct.pack("<ii", len(mass), 1))
            fh.write(np.asarray(mass, "<f8").tobytes())
    snap.read = lambda: None                      # data is already in memory (no h5py here)
    snap.do_snapshots_precomputations(param.q_standarize, param.q_massWeight)

    comp = object.__new__(posComponents)
    comp.basesType = param.vertPos_bases_type
    comp.pos_snapshots = snap
    comp.numComp = param.vertPos_numComponents
    comp.support = param.q_support
    comp.storeSingVal = param.store_vertPos_PCA_sing_val
    comp.comps = comp.weigs = comp.ortho_comps = None
    comp.smooth_min_dist = param.vertPos_smooth_min_dist
    comp.smooth_max_dist = param.vertPos_smooth_max_dist
    comp.output_components_file = "components.h5"
    comp.measures_at_largeDeforVerts = None
    comp.fileNameBases = "q_pos_"
    comp.param = param

    out = dict(snapTensor=snap.snapTensor.copy(), mean=snap.mean.copy(),
               pre_scale_factor=np.float64(snap.pre_scale_factor))
    if mass is not None:
        out.update(mass=snap.mass.copy(), massL=np.array(snap.massL), invMassL=np.array(snap.invMassL))

    # --- observers (pass-through) ---
    picked = []
    real_argmax = pc_mod.argmax

    def rec_argmax(a, *args, **kw):
        r = real_argmax(a, *args, **kw)
        picked.append(int(r))
        return r

    geo_calls = []
    real_geo = snap.compute_geodesic_distance

    def rec_geo(idx):
        phi = real_geo(idx)
        geo_calls.append((int(idx), phi.copy()))
        return phi

    snap.compute_geodesic_distance = rec_geo
    prox_calls, pw_calls = [], []
    if param.vertPos_bases_type == "SPLOCS":
        real_prox, real_pw = posComponents.prox_l1l2, posComponents.project_weight

        def rec_prox(Lambda, x, beta):
            z = real_prox(Lambda, x, beta)
            prox_calls.append((Lambda.copy(), z.copy()))
            return z

        comp.prox_l1l2 = rec_prox
    pc_mod.argmax = rec_argmax
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            comp.compute_components_store_singvalues()
    finally:
        pc_mod.argmax = real_argmax
    K = param.vertPos_numComponents
    out.update(comps=comp.comps.copy(), weigs=comp.weigs.copy(),
               measures=comp.measures_at_largeDeforVerts.copy(),
               idx=np.array(picked[:K], dtype=np.int64))
    csv_path = os.path.join(param.vertPos_output_directory,
                            param.name + "_posBases_pcaExtraction_singValues_errorNorm.csv")
    if param.store_vertPos_PCA_sing_val:
        out["csv_text"] = np.array(open(csv_path).read())
    if param.q_support == "local" or param.vertPos_bases_type == "SPLOCS":
        out["geo_idx"] = np.array([g[0] for g in geo_calls], dtype=np.int64)
        out["geo_phi_first"] = geo_calls[0][1]
        out["geo_phi_deflation"] = np.array([g[1] for g in geo_calls[:K]]) if param.q_support == "local" else np.zeros(0)
    if param.vertPos_bases_type == "SPLOCS":
        trace = []
        for line in buf.getvalue().splitlines():
            if line.startswith("itr "):
                parts = line.replace(",", " ").replace("=", " ").split()
                trace.append([float(parts[3]), float(parts[5])])
        out["splocs_trace"] = np.array(trace)
        admm = param.splocs_admm_num_itrs
        out["splocs_C_final"] = prox_calls[-1][1]
        out["splocs_Lambda_final"] = prox_calls[-1][0]
        out["splocs_C_per_iter"] = np.array([prox_calls[(i + 1) * admm - 1][1] for i in range(param.splocs_max_itrs)])
        out["splocs_centres"] = out["geo_idx"][K:].reshape(param.splocs_max_itrs, K)

    # post-processing (every flag combination asked for by the case) and storage
    with contextlib.redirect_stdout(io.StringIO()):
        comp.post_process_components()
    out["comps_post"] = comp.comps.copy()
    with contextlib.redirect_stdout(io.StringIO()):
        out["bases_sing_vals"] = comp.test_basesSingVals()
        comp.store_components_to_files(K, K, 1, ".bin")
        comp.store_components_to_files(K, K, 1, ".npy")
    bin_path = os.path.join(param.vertPos_output_directory, "q_pos_F%dK%d.bin" % (snap.frs, K))
    npy_path = os.path.join(param.vertPos_output_directory, "q_pos_%dK%d.npy" % (snap.frs, K))
    raw = open(bin_path, "rb").read()
    out["bin_bytes"] = np.frombuffer(raw, dtype=np.uint8)
    out["bin_sha256"] = np.array(hashlib.sha256(raw).hexdigest())
    assert np.array_equal(np.load(npy_path), comp.comps)
    out["bin_name"] = np.array(os.path.basename(bin_path))
    out["npy_name"] = np.array(os.path.basename(npy_path))
    return out


def run_reference_constraints(ref, frames, K, rest_shape, standarize, orthogonal, p, workdir):
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    param = types.SimpleNamespace(constProj_standarize=standarize, constProj_massWeight=False,
                                  constProj_orthogonal=orthogonal, deim_desired_num_components=K,
                                  constProj_output_directory=workdir)
    ns = object.__new__(nonlinearSnapshots)
    ns.param = param
    ns.rest_shape = rest_shape
    ns.dim = 3
    ns.frs = frames.shape[0]
    ns.constraintsSize = p
    ns.num_constained_elements = frames.shape[1] // p
    ns.snapTensor = frames.astype(float).copy()
    ns.mean = None
    ns.pre_scale_factor = 1
    ns.massL = ns.invMassL = None
    if standarize:
        ns.standarize()
    cc = object.__new__(constraintsComponents)
    cc.param = param
    cc.nonlinearSnapshots = ns
    cc.numComp = 0
    cc.comps = None
    cc.geom_interpol_verts = []
    out = dict(snapTensor=ns.snapTensor.copy(),
               mean=ns.mean.copy() if ns.mean is not None else np.zeros(0),
               pre_scale_factor=np.float64(ns.pre_scale_factor))
    rows = []
    writer = types.SimpleNamespace(writerow=lambda r: rows.append(list(r)))
    with contextlib.redirect_stdout(io.StringIO()):
        cc.compute_pod_for_vectorized_nonlinear_snapshots_tensor(writer)
    out["S"] = np.array([r[1] for r in rows])
    out["comps"] = cc.comps.copy()
    with contextlib.redirect_stdout(io.StringIO()):
        cc.post_process_components()
    out["comps_post"] = cc.comps.copy()
    out["snapTensor_post"] = ns.snapTensor.copy()
    with contextlib.redirect_stdout(io.StringIO()):
        cc.deim()
    out["Pt"] = np.asarray(cc.geom_Pt, dtype=np.int64)
    out["alpha"] = np.asarray(cc.geom_alpha, dtype=np.int64)
    out["alpha_ranges"] = np.asarray(cc.geom_alpha_ranges, dtype=np.int64)
    return out


def run_reference_blocks(ref, frames, K, p, standarize=True):
    """constraintsComponents.compute_nonlinearity_bases_blocks (constProj_basis_type 'pca_blocks', :324-412)."""
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    param = types.SimpleNamespace(constProj_standarize=standarize, constProj_massWeight=False, constProj_orthogonal=False,
                                  deim_desired_num_components=K, constProj_output_directory=".")
    ns = object.__new__(nonlinearSnapshots)
    ns.param = param
    ns.rest_shape = "first"
    ns.dim = 3
    ns.frs = frames.shape[0]
    ns.constraintsSize = p
    ns.num_constained_elements = frames.shape[1] // p
    ns.snapTensor = frames.astype(float).copy()
    ns.mean = None
    ns.pre_scale_factor = 1
    ns.massL = ns.invMassL = None
    if standarize:
        ns.standarize()
    cc = object.__new__(constraintsComponents)
    cc.param = param
    cc.nonlinearSnapshots = ns
    cc.numComp = 0
    cc.comps = None
    cc.support = "global"
    cc.storeSingVal = False
    with contextlib.redirect_stdout(io.StringIO()):
        cc.compute_nonlinearity_bases_blocks(None)
    return dict(snapTensor=ns.snapTensor.copy(), comps=cc.comps.copy(), weigs=cc.weigs.copy(),
                measures=cc.measures_at_largeDeforVerts.copy(), points=np.asarray(cc.largeDeforPoints, dtype=np.int64),
                blocks=np.asarray(cc.largeDeforBlocks, dtype=np.int64), numComp=np.array(cc.numComp))


def _blocksdeim(work):
    """Block interpolation on a 'pca_blocks' basis (p = 3): deim_blocksForm (:733-795), geom_block_form_utilizing_
    differential_operator in the constraint space (:619-731), geom_constructed (:489-521) and the files of
    store_components_gradually_to_files (:572-594).  Input: the frames of tests/golden/pca_blocks_p3.npz."""
    import snapbases.constraintsComponents as ccmod
    g = np.load(os.path.join(OUT, "pca_blocks_p3.npz"))
    frames, K, p = g["frames"], int(g["K"]), int(g["p"])
    ref = import_reference.cache
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    out = {}
    for kind in ("deim_block_form", "geom"):
        param = types.SimpleNamespace(constProj_standarize=True, constProj_massWeight=False, constProj_orthogonal=False,
                                      deim_desired_num_components=K, constProj_output_directory=work,
                                      constProj_bases_interpolation_type=kind, constProj_snapshots_type="tris_strain")
        ns = object.__new__(nonlinearSnapshots)
        ns.param, ns.rest_shape, ns.dim, ns.frs, ns.constraintsSize = param, "first", 3, frames.shape[0], p
        ns.num_constained_elements = frames.shape[1] // p
        ns.snapTensor = frames.astype(float).copy()
        ns.mean, ns.pre_scale_factor, ns.massL, ns.invMassL = None, 1, None, None
        ns.standarize()
        cc = object.__new__(constraintsComponents)
        cc.param, cc.nonlinearSnapshots, cc.numComp, cc.comps = param, ns, 0, None
        cc.support, cc.storeSingVal, cc.geom_interpol_verts = "global", False, []
        with contextlib.redirect_stdout(io.StringIO()):
            cc.compute_nonlinearity_bases_blocks(None)
            if kind == "deim_block_form":
                cc.deim_blocksForm()
            else:
                cc.geom_block_form_utilizing_differential_operator(False)
        out[kind + "_Pt"] = np.asarray(cc.geom_Pt, dtype=np.int64)
        out[kind + "_alpha"] = np.asarray(cc.geom_alpha, dtype=np.int64)
        out[kind + "_ranges"] = np.asarray(cc.geom_alpha_ranges, dtype=np.int64)
        # (geom_constructed with p > 1 indexes the ROWS of the basis with constraint numbers, :505: r rows for r p
        # unknowns, a singular normal matrix -- not a parity target; it is pinned with p = 1 in _recon below)
        if kind == "geom":
            out["comps"] = cc.comps.copy()
            out["snapTensor"] = ns.snapTensor.copy()
            ccmod.constProj_output_directory = work
            cc.fileNameBases, cc.fileName_geom_points = "p_nl_", "p_nl_interpol_points_"
            cc.geom_interpol_verts = np.arange(10, 10 + K)
            with contextlib.redirect_stdout(io.StringIO()):
                cc.store_components_gradually_to_files(1, K, 2, ".bin")
            names = sorted(f for f in os.listdir(work) if f.endswith(".bin") and (f.startswith("p_nl_") or f.startswith("corrVerts")))
            out["files"] = np.array(names)
            out["files_sha256"] = np.array([hashlib.sha256(open(os.path.join(work, f), "rb").read()).hexdigest() for f in names])
    np.savez_compressed(os.path.join(OUT, "block_deim_p3.npz"), frames=frames, K=np.array(K), p=np.array(p), **out)
    print("wrote block_deim_p3", out["deim_block_form_alpha"].tolist(), out["geom_alpha"].tolist(), list(out["files"]))


def _withst(work):
    """The S^T variants: 'pca_blocks_with_St' (compute_nonlinearity_bases_blocks_utilizing_diffirential_operator,
    constraintsComponents.py:156-271) and geom_block_form_utilizing_differential_operator(error_in_pos_space=True) (:619-731)
    of the UNMODIFIED reference on a small triangle mesh: e = #triangles constraints of p = 2 rows ('_tris'), a synthetic
    sparse weighted operator S^T (|V| x e p: every vertex sees the rows of the triangles around it), 16 frames."""
    from scipy import sparse
    ref = import_reference.cache
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    rng = np.random.default_rng(41)
    rest, tris = orc.synth_mesh(5, 7, seed=41)
    nv, e, p, F = rest.shape[0], tris.shape[0], 2, 16
    ep = e * p
    rows_, cols_, vals_ = [], [], []
    for t, tri in enumerate(tris):
        for v in tri:
            for m in range(p):
                rows_.append(int(v)); cols_.append(t * p + m); vals_.append(rng.uniform(0.2, 1.0) * (1 if rng.random() < 0.7 else -1))
    St = sparse.csr_matrix((vals_, (rows_, cols_)), shape=(nv, ep))
    modes = rng.normal(size=(F, ep, 3))
    coef = rng.normal(size=(F, F)) * (0.75 ** np.arange(F))[None]
    frames = 0.2 + np.tensordot(coef, modes, (1, 0))
    out = dict(frames=frames, tris=tris, rest=rest, p=np.array(p), St_data=St.data, St_indices=St.indices, St_indptr=St.indptr,
               St_shape=np.array(St.shape))

    def make(kind, K=0, **over):
        param = types.SimpleNamespace(constProj_standarize=True, constProj_massWeight=False, constProj_orthogonal=False,
                                      deim_desired_num_components=K, constProj_output_directory=work,
                                      constProj_bases_interpolation_type=kind, constProj_snapshots_type="tris_strain",
                                      bases_R_tol=1e-8, geom_ele_per_vert=2, **over)
        ns = object.__new__(nonlinearSnapshots)
        ns.param, ns.rest_shape, ns.dim, ns.frs, ns.constraintsSize = param, "first", 3, F, p
        ns.num_constained_elements = e
        ns.snapTensor = frames.astype(float).copy()
        ns.mean, ns.pre_scale_factor, ns.massL, ns.invMassL = None, 1, None, None
        ns.ele_type, ns.tris, ns.tets, ns.edges = "_tris", tris, None, None
        ns.standarize()
        cc = object.__new__(constraintsComponents)
        cc.param, cc.nonlinearSnapshots, cc.numComp, cc.comps = param, ns, 0, None
        cc.support, cc.storeSingVal, cc.geom_interpol_verts, cc.St = "global", False, [], St
        return ns, cc

    ns, cc = make("geom")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cc.compute_nonlinearity_bases_blocks_utilizing_diffirential_operator(None)
    out.update(st_comps=cc.comps.copy(), st_weigs=cc.weigs.copy(), st_measures=cc.measures_at_largeDeforVerts.copy(),
               st_numComp=np.array(cc.numComp),
               st_verts=np.array([int(l.split()[1]) for l in buf.getvalue().splitlines() if l.startswith("vert ")]))
    # position-space interpolation on a 'pca_blocks' basis (K = 6 blocks of p rows)
    K = 6
    ns, cc = make("geom", K)
    with contextlib.redirect_stdout(io.StringIO()):
        cc.compute_nonlinearity_bases_blocks(None)
        cc.geom_block_form_utilizing_differential_operator(True)
    out.update(pos_K=np.array(K), pos_comps=cc.comps.copy(), pos_Pt=np.asarray(cc.geom_Pt, dtype=np.int64),
               pos_alpha=np.asarray(cc.geom_alpha, dtype=np.int64), pos_ranges=np.asarray(cc.geom_alpha_ranges, dtype=np.int64),
               pos_interpol_verts=np.asarray(cc.geom_interpol_verts, dtype=np.int64))
    np.savez_compressed(os.path.join(OUT, "with_st_p2.npz"), **out)
    print("wrote with_st_p2: components", out["st_comps"].shape, "verts", out["st_verts"].tolist(), "| pos-space alpha",
          out["pos_alpha"].tolist(), "ranges", out["pos_ranges"].tolist(), "verts", out["pos_interpol_verts"].tolist())


def _podslices(work):
    """constProj_basis_type 'pod' (compute_pod_for_nonlinear_snapshots_tensor, :274-294; torch float32 SVD per (p, d) slice)."""
    ref = import_reference.cache
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    rng = np.random.default_rng(31)
    e, p, F, K = 36, 2, 14, 5
    modes = rng.normal(size=(9, e * p, 3))
    coef = rng.normal(size=(F, 9)) * (0.6 ** np.arange(9))[None]
    frames = 0.3 + np.tensordot(coef, modes, (1, 0)) + 1e-4 * rng.normal(size=(F, e * p, 3))
    param = types.SimpleNamespace(constProj_standarize=True, constProj_massWeight=False, constProj_orthogonal=False,
                                  deim_desired_num_components=K, constProj_output_directory=work)
    ns = object.__new__(nonlinearSnapshots)
    ns.param, ns.rest_shape, ns.dim, ns.frs, ns.constraintsSize = param, "first", 3, F, p
    ns.num_constained_elements = e
    ns.snapTensor = frames.astype(float).copy()
    ns.mean, ns.pre_scale_factor, ns.massL, ns.invMassL = None, 1, None, None
    ns.standarize()
    cc = object.__new__(constraintsComponents)
    cc.param, cc.nonlinearSnapshots, cc.numComp, cc.comps = param, ns, 0, None
    with contextlib.redirect_stdout(io.StringIO()):
        cc.compute_pod_for_nonlinear_snapshots_tensor(None)
    np.savez_compressed(os.path.join(OUT, "pod_slices_p2.npz"), frames=frames, K=np.array(K), p=np.array(p),
                        comps=np.asarray(cc.comps, dtype=np.float64), numComp=np.array(cc.numComp))
    print("wrote pod_slices_p2", cc.comps.shape, cc.comps.dtype)


def _recon(work):
    """geom_constructed (:489-521) after deim on the POD basis of tests/golden/pod_deim_small.npz (p = 1), train and test frames."""
    g = np.load(os.path.join(OUT, "pod_deim_small.npz"))
    frames, K = g["frames"], int(g["K"])
    ref = import_reference.cache
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    param = types.SimpleNamespace(constProj_standarize=True, constProj_massWeight=False, constProj_orthogonal=False,
                                  deim_desired_num_components=K, constProj_output_directory=work,
                                  constProj_bases_interpolation_type="deim", constProj_snapshots_type="tris_strain")
    ns = object.__new__(nonlinearSnapshots)
    ns.param, ns.rest_shape, ns.dim, ns.frs, ns.constraintsSize = param, "first", 3, frames.shape[0], 1
    ns.num_constained_elements = frames.shape[1]
    ns.snapTensor = frames.astype(float).copy()
    ns.mean, ns.pre_scale_factor, ns.massL, ns.invMassL = None, 1, None, None
    ns.standarize()
    rng = np.random.default_rng(77)
    ns.test_snapTensor = ns.snapTensor[::3] + 1e-3 * rng.normal(size=ns.snapTensor[::3].shape)
    cc = object.__new__(constraintsComponents)
    cc.param, cc.nonlinearSnapshots, cc.numComp, cc.comps, cc.geom_interpol_verts = param, ns, 0, None, []
    out = dict(test_snapTensor=ns.test_snapTensor.copy())
    with contextlib.redirect_stdout(io.StringIO()):
        cc.compute_pod_for_vectorized_nonlinear_snapshots_tensor(None)
        cc.deim()
    out["Pt"] = np.asarray(cc.geom_Pt, dtype=np.int64)
    for r in (3, K):
        out["train_r%d" % r] = cc.geom_constructed(r, "train")
        out["test_r%d" % r] = cc.geom_constructed(r, "test")
    np.savez_compressed(os.path.join(OUT, "pod_deim_recon.npz"), K=np.array(K), **out)
    print("wrote pod_deim_recon", out["Pt"].tolist())


def _blocks(work):
    rng = np.random.default_rng(21)
    e, F, K = 40, 20, 5
    for p in (1, 3):
        modes = rng.normal(size=(8, e * p, 3))
        coef = rng.normal(size=(F, 8)) * (0.7 ** np.arange(8))[None]
        frames = 0.2 + np.tensordot(coef, modes, (1, 0)) + 1e-4 * rng.normal(size=(F, e * p, 3))
        res = run_reference_blocks(import_reference.cache, frames, K, p)
        np.savez_compressed(os.path.join(OUT, "pca_blocks_p%d.npz" % p), frames=frames, K=np.array(K), p=np.array(p), **res)
        print("wrote pca_blocks_p%d" % p, "points", res["points"].tolist())


CASES = {
    # name: (rings, segs, F, rank, kind, seed, param overrides, with_mass)
    "pca_global_small": (8, 12, 40, 5, "iid", 1, dict(vertPos_numComponents=6), False),
    "pca_global_avg_mass_orth": (8, 12, 36, 6, "iid", 2,
                                 dict(vertPos_numComponents=5, vertPos_rest_shape="average", q_massWeight=True,
                                      q_orthogonal=True), True),
    "pca_global_nostd": (6, 10, 24, 4, "iid", 3, dict(vertPos_numComponents=4, q_standarize=False), False),
    "pca_global_medium": (16, 30, 64, 12, "iid", 4, dict(vertPos_numComponents=16), False),
    "pca_local_small": (10, 16, 30, 5, "bumps", 5,
                        dict(vertPos_numComponents=5, q_support="local", vertPos_smooth_min_dist=0.1,
                             vertPos_smooth_max_dist=0.35), False),
    "splocs_small": (10, 16, 30, 5, "bumps", 6,
                     dict(vertPos_numComponents=5, q_support="local", vertPos_bases_type="SPLOCS",
                          vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.35,
                          splocs_max_itrs=3, splocs_admm_num_itrs=4), False),
}


def _ingest(work):
    # ingest: .off loading, pre-processing, rigid Procrustes alignment (utils/process.py)
    import utils.process as rp
    rng = np.random.default_rng(21)
    rest, tris = orc.synth_mesh(6, 9, seed=9)
    extra = rest[:4] * 0.1 + 2.0                         # a small disconnected component (4 vertices, 2 triangles)
    rest2 = np.vstack([rest, extra])
    n0 = rest.shape[0]
    tris2 = np.vstack([tris, [[n0, n0 + 1, n0 + 2], [n0 + 1, n0 + 2, n0 + 3], [0, 0, 1]]])   # + a zero-area triangle
    offdir = os.path.join(work, "off")
    os.makedirs(offdir)
    raw = []
    for f in range(7):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = 0.4 * f
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rm = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        vf = (rest2 + 0.02 * f * rng.normal(size=rest2.shape) * (np.arange(rest2.shape[0]) % 3 == 0)[:, None]) @ Rm.T \
            + rng.normal(size=3)
        raw.append(vf)
        with open(os.path.join(offdir, "frame_%d.off" % f), "w") as fh:
            fh.write("OFF\n# comment line\n%d %d 0\n" % (vf.shape[0], tris2.shape[0]))
            for v in vf:
                fh.write("%.9f %.9f %.9f\n" % tuple(v))
            for t in tris2:
                fh.write("3 %d %d %d\n" % tuple(t))
    loaded = [rp.load_off(os.path.join(offdir, "frame_%d.off" % f), no_colors=True) for f in range(7)]
    verts_all = np.array([l[0] for l in loaded], np.float32)
    class _PtpArray(np.ndarray):      # the reference calls ndarray.ptp (gone in NumPy 2): give the INPUT that method
        def ptp(self, *a, **k):
            return np.ptp(np.asarray(self), *a, **k)

    with contextlib.redirect_stdout(io.StringIO()):
        pv, pt, removed, pmean, pscale = rp.preprocess_mesh_animation(verts_all.copy().view(_PtpArray), loaded[0][1])
    pv, pmean = np.asarray(pv), np.asarray(pmean)
    out = {}
    for rigid in (True, False):
        v0 = pv[0]
        Ts, new = [], []
        for v in pv:
            M = rp.find_rbm_procrustes(v, v0, rigid)
            Ts.append(M)
            new.append(rp.transform(v, M))
        out["aligned_rigid%d" % int(rigid)] = np.array(new, np.float32)
        out["T_rigid%d" % int(rigid)] = np.array(Ts)
    np.savez_compressed(os.path.join(OUT, "ingest_small.npz"), off_verts=np.array([l[0] for l in loaded]),
                        off_tris=loaded[0][1], pre_verts=pv, pre_tris=pt, pre_removed=removed, pre_mean=pmean,
                        pre_scale=np.array(pscale), **out)
    for f in range(7):
        with open(os.path.join(offdir, "frame_%d.off" % f)) as fh:
            pass
    np.savez_compressed(os.path.join(OUT, "ingest_small_off.npz"),
                        **{"frame_%d" % f: np.array(open(os.path.join(offdir, "frame_%d.off" % f)).read()) for f in range(7)})
    print("wrote ingest_small")


def _diagnostics(work):
    """The diagnostics of constraintsComponents (constraintsComponents.py:452-487, 524-570) run by the unmodified reference on the
    small POD + DEIM case: matrix_properties_test on the DEIM points, test_basesSingVals, the printed verdicts of
    is_utmu_orthogonal, the three error metrics on a perturbed copy of the snapshots."""
    ref = import_reference.cache
    constraintsComponents, nonlinearSnapshots = ref["constraintsComponents"], ref["nonlinearSnapshots"]
    rng = np.random.default_rng(11)
    ep, F, K = 120, 24, 8
    modes = rng.normal(size=(10, ep, 3))
    coef = rng.normal(size=(F, 10)) * (0.6 ** np.arange(10))[None]
    frames = 0.3 + np.tensordot(coef, modes, (1, 0)) + 1e-5 * rng.normal(size=(F, ep, 3))          # = the pod_deim_small input
    param = types.SimpleNamespace(constProj_standarize=True, constProj_massWeight=False, constProj_orthogonal=True,
                                  deim_desired_num_components=K, constProj_output_directory=work)
    ns = object.__new__(nonlinearSnapshots)
    ns.param, ns.rest_shape, ns.dim, ns.frs, ns.constraintsSize = param, "first", 3, F, 1
    ns.num_constained_elements = ep
    ns.snapTensor = frames.copy()
    ns.mean, ns.pre_scale_factor, ns.massL, ns.invMassL = None, 1, None, None
    ns.mass = rng.uniform(0.5, 2.0, size=ep)
    ns.standarize()
    cc = object.__new__(constraintsComponents)
    cc.param, cc.nonlinearSnapshots, cc.numComp, cc.comps, cc.geom_interpol_verts = param, ns, 0, None, []
    buf = io.StringIO()
    with contextlib.redirect_stdout(io.StringIO()):
        cc.compute_pod_for_vectorized_nonlinear_snapshots_tensor(types.SimpleNamespace(writerow=lambda r: None))
        cc.post_process_components()
        cc.deim()
    with contextlib.redirect_stdout(buf):
        cc.is_utmu_orthogonal()
    mat_e = cc.matrix_properties_test(np.asarray(cc.geom_Pt))
    sv = cc.test_basesSingVals()
    rec = ns.snapTensor + 1e-3 * rng.normal(size=ns.snapTensor.shape)
    np.savez_compressed(os.path.join(OUT, "constraints_diagnostics.npz"), frames=frames, mass=ns.mass, K=np.array(K),
                        comps=cc.comps.copy(), Pt=np.asarray(cc.geom_Pt, dtype=np.int64), snapTensor=ns.snapTensor.copy(),
                        mat_e=mat_e, bases_sing_vals=sv, utmu_stdout=np.array(buf.getvalue()), rec=rec,
                        frobenius=np.float64(constraintsComponents.frobenius_error(ns.snapTensor, rec)),
                        relative=np.array(constraintsComponents.relative_error_per_component(ns.snapTensor, rec)),
                        max_pointwise=np.float64(constraintsComponents.max_pointwise_error(ns.snapTensor, rec)))
    print("wrote constraints_diagnostics: mat_e", mat_e.shape, "utmu:", buf.getvalue().replace("\n", " | "))


def _meshmass(work):
    """Element masses derived from a mesh -- the reference's OWN arithmetic (utils/support.py:12-76: `compute_lumped_mass_matrix`,
    `compute_tetMasses`, `compute_edgeMasses`, `compute_triMasses`) and its `.mesh` reader (utils/utils.py:325-389) run on a small
    random tetrahedral mesh written to disk here.  (The vertex masses the reference takes from libigl -- `igl.massmatrix` -- are not
    part of this fixture: no libigl in this image.)"""
    from utils.support import compute_lumped_mass_matrix, compute_tetMasses, compute_edgeMasses, compute_triMasses
    ref = import_reference.cache
    rng = np.random.default_rng(31)
    # a 3 x 3 x 2 grid of jittered cubes, five tetrahedra each
    nx, ny, nz = 4, 4, 3
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    V = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(float) + 0.15 * rng.normal(size=(nx * ny * nz, 3))
    vid = lambda i, j, k: (i * ny + j) * nz + k
    T = []
    for i in range(nx - 1):
        for j in range(ny - 1):
            for k in range(nz - 1):
                c = [vid(i + a, j + b, k + d) for a in (0, 1) for b in (0, 1) for d in (0, 1)]     # c[4a + 2b + d]
                T += [[c[0], c[4], c[2], c[1]], [c[6], c[2], c[4], c[7]], [c[5], c[4], c[1], c[7]], [c[3], c[1], c[2], c[7]],
                      [c[4], c[2], c[1], c[7]]]
    T = np.array(T, dtype=np.int64)
    faces = np.sort(np.concatenate([T[:, [0, 1, 2]], T[:, [0, 1, 3]], T[:, [0, 2, 3]], T[:, [1, 2, 3]]]), axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    tris = uniq[counts == 1]                                   # boundary triangles
    mesh_path = os.path.join(work, "cubes.mesh")
    with open(mesh_path, "w") as fh:
        fh.write("MeshVersionFormatted 1\nDimension 3\nVertices\n%d\n" % len(V))
        for p in V:
            fh.write("%.17g %.17g %.17g 0\n" % tuple(p))
        fh.write("Tetrahedra\n%d\n" % len(T))
        for t in T:
            fh.write("%d %d %d %d 0\n" % tuple(t + 1))
        fh.write("Triangles\n%d\n" % len(tris))
        for t in tris:
            fh.write("%d %d %d 0\n" % tuple(t + 1))
        fh.write("End\n")
    rV, rT, rTri = ref["utils"].read_mesh_file(mesh_path)
    vm = np.asarray(compute_lumped_mass_matrix(rV, rT).todense()).diagonal().copy()
    edges = np.unique(np.sort(np.concatenate([rT[:, [a, b]] for a in range(4) for b in range(a + 1, 4)]), axis=1), axis=0)
    out = dict(mesh_text=np.frombuffer(open(mesh_path, "rb").read(), dtype=np.uint8), read_V=rV, read_T=rT, read_tris=rTri,
               lumped_vertex_mass=vm, edges=edges,
               tet_masses=compute_tetMasses(vm, rT, rT.shape[0], 3),
               edge_masses=compute_edgeMasses(vm, edges, edges.shape[0], 1),
               tri_masses=compute_triMasses(vm, rTri, rTri.shape[0], 2))
    np.savez_compressed(os.path.join(OUT, "mesh_masses.npz"), **out)
    print("wrote mesh_masses: %d vertices, %d tets, %d boundary triangles, %d edges" % (len(rV), len(rT), len(rTri), len(edges)))


def main():
    """`python oracle/gen_golden.py` regenerates every fixture; `... ingest` / `... blocks` only that one.  The ingest
    functions of the reference (utils/process.py) call `ndarray.ptp` / `np.asfarray`, which NumPy 2 removed, so that
    fixture is generated with the image's other interpreter: `/opt/conda/bin/python3.9 oracle/gen_golden.py ingest`
    (NumPy 1.26)."""
    only = sys.argv[1:] or None
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    import_reference.cache = ref
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as work:
        os.chdir(work)                             # log_time writes function_timings.txt into cwd
        try:
            for name, (rings, segs, F, rank, kind, seed, over, with_mass) in (CASES.items() if only is None else []):
                rest, tris = orc.synth_mesh(rings, segs, seed=seed)
                noise = 1e-4 if "nostd" not in name else 1e-3
                verts = orc.synth_snapshots(rest, F, rank=rank, noise=noise, seed=seed, kind=kind)
                mass = None
                if with_mass:
                    mass = np.random.default_rng(seed + 100).uniform(0.5, 2.0, size=rest.shape[0])
                    mass = mass / mass.sum() * 2
                param = _param(vertPos_output_directory=work, **over)
                res = run_reference_pos(ref, verts, tris, param, mass=mass, workdir=work)
                if "medium" in name:
                    res.pop("snapTensor")          # keep the fixture small; it is recomputed from verts
                meta = dict(rings=rings, segs=segs, F=F, rank=rank, seed=seed, noise=noise)
                np.savez_compressed(os.path.join(OUT, name + ".npz"), verts=verts, tris=tris,
                                    kind=np.array(kind),
                                    **{"meta_" + k: np.array(v) for k, v in meta.items()},
                                    **{"param_" + k: np.array(v) for k, v in vars(param).items()
                                       if k != "vertPos_output_directory"},
                                    **res)
                print("wrote", name, "idx", res["idx"].tolist())
            if only is not None and "meshmass" in only:
                return _meshmass(work)
            if only is not None and "diagnostics" in only:
                return _diagnostics(work)
            if only is not None and "blocksdeim" in only:
                return _blocksdeim(work)
            if only is not None and "recon" in only:
                return _recon(work)
            if only is not None and "podslices" in only:
                return _podslices(work)
            if only is not None and "withst" in only:
                return _withst(work)
            if only is not None and "blocks" in only:
                return _blocks(work)
            if only is not None and "ingest" not in only:
                return
            if only is not None:
                return _ingest(work)
            # config-5 style: POD + DEIM on constraint-projection snapshots
            rng = np.random.default_rng(11)
            ep, F, K = 120, 24, 8
            modes = rng.normal(size=(10, ep, 3))
            coef = rng.normal(size=(F, 10)) * (0.6 ** np.arange(10))[None]
            frames = 0.3 + np.tensordot(coef, modes, (1, 0)) + 1e-5 * rng.normal(size=(F, ep, 3))
            for nm, orth in (("pod_deim_small", False), ("pod_deim_small_qr", True)):
                res = run_reference_constraints(ref, frames, K, "first", True, orth, 1, work)
                np.savez_compressed(os.path.join(OUT, nm + ".npz"), frames=frames, K=np.array(K),
                                    p=np.array(1), orthogonal=np.array(orth), **res)
                print("wrote", nm, "Pt", res["Pt"].tolist())
            _blocks(work)
            _blocksdeim(work)
            _recon(work)
            _podslices(work)
            _withst(work)
            _meshmass(work)
            _diagnostics(work)
            if np.lib.NumpyVersion(np.__version__) < '2.0.0':
                _ingest(work)
        finally:
            os.chdir(cwd)


if __name__ == "__main__":
    main()
