"""GPU (-m gpu): the HIP path, called through the C ABI behind the posSnapshots /
posComponents mirror, against (a) golden vectors produced by the unmodified reference and
(b) the NumPy oracle on seeded inputs.

Bars (BASELINE.json north_star): selected-vertex index sequence bit-exact; basis / weights
/ singular values within 1e-5 relative Frobenius error -- the tests hold the HIP path to
1e-9, far inside that.  Global-support components carry LAPACK's arbitrary SVD sign and are
compared after per-component sign alignment.
"""
import types

import numpy as np
import pytest

from conftest import align_signs, load_golden, relerr
from oracle import asb_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _param(g=None, **over):
    base = dict(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                q_support="global", vertPos_numComponents=4, store_vertPos_PCA_sing_val=False,
                vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, vertPos_rest_shape="first",
                name="t", vertPos_output_directory=".")
    if g is not None:
        for k in list(base):
            if "param_" + k in g:
                v = g["param_" + k]
                base[k] = v.item() if v.ndim == 0 else v
                if isinstance(base[k], (np.str_, bytes)):
                    base[k] = str(base[k])
    base.update(over)
    return types.SimpleNamespace(**base)


def _run(verts, tris, param, mass=None, mode=None):
    from animsnapbases_amd import posComponents, posSnapshots
    snaps = posSnapshots.from_arrays(verts, tris, param.vertPos_rest_shape, standarize=param.q_standarize,
                                     massWeight=param.q_massWeight, mass=mass)
    comp = posComponents(param, snaps)
    comp.deflate_mode = mode
    comp.compute_components_store_singvalues()
    return snaps, comp


MODES = ["residual", "project"]     # both device algorithms for support='global'


def test_library_is_the_hip_one():
    from animsnapbases_amd import HipEngine
    e = HipEngine(0)
    assert e.lib.asb_abi_version() == 1
    e.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["pca_global_small", "pca_global_nostd", "pca_global_medium",
                                  "pca_global_avg_mass_orth"])
def test_global_deflation_vs_reference_golden(name, mode, tmp_path):
    g = load_golden(name)
    param = _param(g, q_orthogonal=False, store_vertPos_PCA_sing_val=True, vertPos_output_directory=str(tmp_path))
    mass = g["mass"] if bool(g["param_q_massWeight"]) else None
    snaps, comp = _run(g["verts"], g["tris"], param, mass, mode)
    assert relerr(snaps.mean, g["mean"]) < 1e-13
    assert abs(snaps.pre_scale_factor - float(g["pre_scale_factor"])) < 1e-12 * float(g["pre_scale_factor"])
    if "snapTensor" in g:
        assert relerr(snaps.snapTensor, g["snapTensor"]) < 1e-13
    assert comp.selected_vertices.tolist() == g["idx"].tolist()          # bit-exact index selection
    comps, weigs = align_signs(comp.comps, comp.weigs, g["comps"])
    assert relerr(comps, g["comps"]) < TOL
    assert relerr(weigs, g["weigs"]) < TOL
    # project mode gets ||R|| from |X|^2 - sum |w|^2|c|^2 (cancellation once the residual is ~1e-4 of X)
    mtol = TOL if mode == "residual" else 1e-7
    assert relerr(comp.measures_at_largeDeforVerts, g["measures"]) < mtol
    # CSV: same header and rows as the reference's file
    lines = open(str(tmp_path / (param.name + "_posBases_pcaExtraction_singValues_errorNorm.csv"))).read().splitlines()
    assert lines[0] == "component,singVal,norm_R"
    rows = np.array([[float(x) for x in ln.split(",")] for ln in lines[1:] if ln])
    assert relerr(rows, g["measures"]) < mtol


@pytest.mark.parametrize("name", ["pca_global_small", "pca_global_nostd"])
def test_post_process_and_bin_vs_reference_golden(name, tmp_path):
    g = load_golden(name)
    param = _param(g, vertPos_output_directory=str(tmp_path))
    snaps, comp = _run(g["verts"], g["tris"], param)
    signs = np.array([1.0 if np.vdot(comp.comps[k], g["comps"][k]) >= 0 else -1.0 for k in range(comp.numComp)])
    comp.post_process_components()
    # a sign flip of comps[k] before "+ mean" is not a flip afterwards: undo it in the standardised space
    std = bool(g["param_q_standarize"])
    got = comp.comps.copy()
    if std:
        got = (got - snaps.mean[None]) * signs[:, None, None] + snaps.mean[None]
    else:
        got = got * signs[:, None, None]
    assert relerr(got, g["comps_post"]) < TOL
    K = comp.numComp
    comp.store_components_to_files(K, K, 1, ".bin")
    from animsnapbases_amd.utils import read_components_bin
    back = read_components_bin(str(tmp_path / str(g["bin_name"])))
    assert np.array_equal(back, comp.comps)


def test_local_support_vs_reference_golden():
    g = load_golden("pca_local_small")
    param = _param(g)
    snaps, comp = _run(g["verts"], g["tris"], param)
    assert comp.selected_vertices.tolist() == g["idx"].tolist()
    # local support canonicalises the sign through the +-projection test: no alignment
    assert relerr(comp.weigs, g["weigs"]) < TOL
    assert relerr(comp.comps, g["comps"]) < 1e-8
    assert relerr(comp.measures_at_largeDeforVerts, g["measures"]) < 1e-8
    assert (comp.weigs >= 0).all() and np.allclose(comp.weigs.max(axis=0), 1.0)


@pytest.mark.parametrize("rings,segs,F,K,rest,seed", [
    (5, 7, 13, 3, "first", 0),        # ragged: N = 37, F = 13 (not multiples of anything)
    (9, 11, 200, 8, "average", 1),    # F = 200 -> one wave per vertex
    (4, 6, 1000, 6, "first", 2),      # F = 1000 -> 128 threads per vertex
    (3, 5, 2000, 5, "first", 3),      # F = 2000 -> 256 threads per vertex (config-4 row length)
    (3, 4, 4100, 4, "first", 4),      # 1024 threads per vertex
    (2, 3, 9000, 3, "first", 5),      # E2 = 8 variant
])
@pytest.mark.parametrize("mode", MODES)
def test_global_deflation_vs_oracle_shapes(rings, segs, F, K, rest, seed, mode):
    rest_v, tris = orc.synth_mesh(rings, segs, seed=seed)
    verts = orc.synth_snapshots(rest_v, F, rank=6, seed=seed)
    param = _param(vertPos_numComponents=K, vertPos_rest_shape=rest)
    snaps, comp = _run(verts, tris, param, mode=mode)
    pre = orc.prepare_snapshots(verts, rest, True)
    ref = orc.extract_k_components(pre["snapTensor"], K)
    assert relerr(snaps.snapTensor, pre["snapTensor"]) < 1e-12
    assert comp.selected_vertices.tolist() == ref["idx"].tolist()
    comps, weigs = align_signs(comp.comps, comp.weigs, ref["comps"])
    assert relerr(comps, ref["comps"]) < TOL
    assert relerr(weigs, ref["weigs"]) < TOL
    assert relerr(comp.measures_at_largeDeforVerts, ref["measures"]) < (TOL if mode == "residual" else 1e-7)
    if mode == "residual":          # residual after K components equals the oracle's
        R = snaps._engine.download_residual()
        assert relerr(R, ref["R"]) < 1e-8


def test_stepwise_equals_fused_and_ties_pick_first():
    """pick/apply step by step == run_global; and with exactly tied energies the FIRST vertex wins
    (NumPy argmax tie-break, posComponents.py:79)."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(5)
    F, N = 24, 50
    X = rng.normal(size=(F, N, 3))
    X[:, 31] = X[:, 7]          # identical trajectories -> identical energies: vertex 7 must win over 31
    X[:, 7] *= 3
    X[:, 31] *= 3
    outs = []
    for fused in (True, False):
        e = HipEngine(0)
        e.upload(X, 0, N)
        e.deflate_begin(4, False)
        if fused:
            e.run_global(0, 4)
        else:
            for k in range(4):
                e.pick(k)
                e.apply(k)
        outs.append(e.results())
        e.close()
    assert outs[0]["idx"].tolist() == outs[1]["idx"].tolist()
    assert np.array_equal(outs[0]["comps"], outs[1]["comps"])
    assert outs[0]["idx"][0] == 7
    ref = orc.extract_k_components(X, 4)
    assert outs[0]["idx"].tolist() == ref["idx"].tolist()


def test_global_properties_full_row_length():
    """Size-independent properties at a config-4-like row length (F = 2000) on a mid-size shard:
    W columns mutually orthogonal, X = W C^T + R, ||R|| as recorded, energies non-increasing."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(11)
    F, N, K = 2000, 3000, 12
    X = rng.uniform(-1, 1, size=(F, N, 3))
    e = HipEngine(0)
    e.upload(X, 0, N)
    e.deflate_begin(K, False)
    e.run_global(0, K)
    r = e.results()
    R = e.download_residual()
    e.close()
    W, C = r["weigs"], r["comps"]
    G = W.T @ W
    off = G - np.diag(np.diag(G))
    assert np.abs(off).max() < 1e-9 * np.diag(G).min()
    rec = np.tensordot(W, C, (1, 0)) + R
    assert relerr(rec, X) < 1e-13
    assert abs(np.sqrt(r["normR2_local"][-1]) - np.linalg.norm(R)) < 1e-10 * np.linalg.norm(R)
    assert (np.diff(r["normR2_local"]) <= 0).all()
    # the selected vertex really had the largest residual energy at step 0
    assert r["idx"][0] == int(np.argmax((X ** 2).sum(axis=(0, 2))))
    assert len(set(r["idx"].tolist())) == K


@pytest.mark.parametrize("kind,N,F,K", [("random", 5000, 200, 40), ("lowrank", 6000, 120, 24)])
def test_project_mode_many_panels(kind, N, F, K):
    """N above the candidate capacity: threshold selection, several panels, early panel ends.
    Projection mode == residual mode == oracle (index sequence exact)."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(21)
    if kind == "random":
        X = rng.uniform(-1, 1, size=(F, N, 3))
    else:
        X = np.tensordot(rng.normal(size=(F, 8)) * (0.7 ** np.arange(8)), rng.normal(size=(8, N, 3)), (1, 0)) \
            + 1e-4 * rng.normal(size=(F, N, 3))
    outs = {}
    for mode in (0, 1):
        e = HipEngine(0)
        e.upload(X, 0, N)
        e.deflate_begin(K, False, mode)
        e.run_global(0, K)
        outs[mode] = e.results()
        if mode == 1:
            st = e.deflate_stats()
            assert 1 <= st["panels"] <= K and st["refreshes"] <= 2, st
        e.close()
    ref = orc.extract_k_components(X, K)
    for mode in (0, 1):
        assert outs[mode]["idx"].tolist() == ref["idx"].tolist(), mode
        comps, weigs = align_signs(outs[mode]["comps"], outs[mode]["weigs"], ref["comps"])
        assert relerr(comps, ref["comps"]) < 1e-8
        assert relerr(weigs, ref["weigs"]) < 1e-8
        assert relerr(outs[mode]["sigma"], ref["measures"][:, 1]) < 1e-9
        assert relerr(np.sqrt(outs[mode]["normR2_local"]), ref["measures"][:, 2]) < 1e-6


def test_splocs_vs_reference_golden(capsys):
    """SPLOCS refinement on the GPU against the trace / refined components the unmodified reference
    produced (captured through pass-through wrappers, oracle/gen_golden.py)."""
    g = load_golden("splocs_small")
    param = _param(g, splocs_max_itrs=int(g["param_splocs_max_itrs"]),
                   splocs_admm_num_itrs=int(g["param_splocs_admm_num_itrs"]),
                   splocs_lambda=float(g["param_splocs_lambda"]), splocs_rho=float(g["param_splocs_rho"]))
    snaps, comp = _run(g["verts"], g["tris"], param)
    # fact 2 of SURVEY.md: comps / weigs are those of PCA-local, untouched by SPLOCS
    assert comp.selected_vertices.tolist() == g["idx"].tolist()
    assert relerr(comp.comps, g["comps"]) < 1e-8
    assert relerr(comp.weigs, g["weigs"]) < TOL
    # the reference prints %f (6 decimals): compare parsed floats
    assert np.allclose(comp.splocs_trace, g["splocs_trace"], rtol=1e-8, atol=2e-6)
    assert comp.splocs_centres.tolist() == g["splocs_centres"].tolist()
    assert relerr(comp.splocs_comps, g["splocs_C_final"]) < 1e-8
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.startswith("itr ")]
    assert len(lines) == int(g["param_splocs_max_itrs"]) and lines[0].startswith("itr 000, Energy =")


def test_splocs_vs_oracle_medium():
    """A larger SPLOCS run (K = 12, 4 outer x 5 ADMM) against the NumPy oracle."""
    rest_v, tris = orc.synth_mesh(14, 20, seed=8)          # N = 282
    verts = orc.synth_snapshots(rest_v, 60, rank=8, seed=8, kind="bumps")
    K = 12
    param = _param(vertPos_numComponents=K, q_support="local", vertPos_bases_type="SPLOCS",
                   vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.4, splocs_max_itrs=4,
                   splocs_admm_num_itrs=5, splocs_lambda=2.0, splocs_rho=10.0)
    snaps, comp = _run(verts, tris, param)
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris)
    d = orc.extract_k_components(pre["snapTensor"], K, "local", geo, 0.1, 0.4)
    s = orc.splocs_glob_optimization(pre["snapTensor"], d["comps"], d["weigs"], d["R"], geo, 0.1, 0.4, 4, 5, 2.0, 10.0)
    assert comp.selected_vertices.tolist() == d["idx"].tolist()
    assert comp.splocs_centres.tolist() == s["idx"].tolist()
    assert np.allclose(comp.splocs_trace, s["trace"], rtol=1e-8)
    assert relerr(comp.splocs_comps, s["C"]) < 1e-8
    assert relerr(comp.splocs_weigs, s["W"]) < 1e-8


def test_splocs_weight_sweep_blocks_agree(monkeypatch):
    """The activation sweep spread over co-resident blocks (k_bcd_wide: rows in LDS, one exchange of the column maximum
    per column) must give the single-block kernel's weights bit for bit (F = 700: three blocks, one of them ragged)."""
    rest_v, tris = orc.synth_mesh(14, 20, seed=18)
    verts = orc.synth_snapshots(rest_v, 700, rank=20, seed=18, kind="bumps")
    param = _param(vertPos_numComponents=24, q_support="local", vertPos_bases_type="SPLOCS", vertPos_smooth_min_dist=0.1,
                   vertPos_smooth_max_dist=0.4, splocs_max_itrs=3, splocs_admm_num_itrs=4, splocs_lambda=2.0, splocs_rho=10.0)
    outs = []
    for wide in ("1", "0"):
        monkeypatch.setenv("ASB_BCD_WIDE", wide)
        snaps, comp = _run(verts, tris, param)
        outs.append((comp.splocs_weigs.copy(), comp.splocs_comps.copy(), comp.splocs_trace.copy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])
    assert outs[0][0].max() == 1.0 and outs[0][0].min() == 0.0


def test_unproven_panel_steps_keep_the_reference_sequence(monkeypatch):
    """Panels may append steps whose winner cannot be proven in advance; the projection pass checks them against every
    vertex's energy and keeps the valid prefix.  On data where the provable panels are short (a common direction in
    every vertex: uniform noise with rest_shape='first') this must save passes and still give the oracle's sequence;
    with ASB_SPEC_PANELS=0 the same results come out of more panels."""
    rng = np.random.default_rng(71)
    F, N, K = 160, 30000, 40
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    param = _param(vertPos_numComponents=K)
    from animsnapbases_amd import posComponents, posSnapshots
    from animsnapbases_amd import HipEngine
    outs = []
    for spec, stepwise in (("1", False), ("0", False), ("1", True)):      # third: the multi-rank protocol's form of it
        monkeypatch.setenv("ASB_SPEC_PANELS", spec)
        eng = HipEngine(0, stream=0) if stepwise else None
        snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False, engine=eng)
        comp = posComponents(param, snaps)
        comp.deflate_mode = "project"
        comp._stepwise_panels = stepwise
        comp.compute_components_store_singvalues()
        outs.append((comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(),
                     comp.measures_at_largeDeforVerts.copy(), snaps._engine.deflate_stats()))
    pre = orc.prepare_snapshots(verts, "first", True)
    d = orc.extract_k_components(pre["snapTensor"], K, "global", None, 0.1, 0.4)
    for o in outs:
        assert o[0].tolist() == d["idx"].tolist()
        comps, weigs = align_signs(o[1], o[2], d["comps"])
        assert relerr(comps, d["comps"]) < 1e-9 and relerr(weigs, d["weigs"]) < 1e-9
        assert relerr(o[3][:, 1:], d["measures"][:, 1:]) < 1e-8
    st1, st0, st2 = outs[0][4], outs[1][4], outs[2][4]
    assert st0["unproven_tried"] == 0 and st0["unproven_kept"] == 0
    assert st1["unproven_kept"] > 0 and st1["panels"] < st0["panels"], (st1, st0)
    assert st2["unproven_kept"] > 0 and st2["panels"] < st0["panels"], (st2, st0)
    assert relerr(outs[0][1], outs[1][1]) < 1e-11 and relerr(outs[0][2], outs[1][2]) < 1e-11
    assert relerr(outs[2][1], outs[1][1]) < 1e-11 and relerr(outs[2][2], outs[1][2]) < 1e-11


def test_double_panels_keep_the_reference_sequence(monkeypatch):
    """ASB_DOUBLE_PANELS=1 (experimental): two sub-panels on the same candidate rows, one 32-column read of X, the
    unproven steps checked tile by tile -- fewer reads of X, same vertex sequence and basis as the oracle."""
    rng = np.random.default_rng(72)
    F, N, K = 144, 26000, 52
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    param = _param(vertPos_numComponents=K)
    from animsnapbases_amd import posComponents, posSnapshots
    pre = orc.prepare_snapshots(verts, "first", True)
    d = orc.extract_k_components(pre["snapTensor"], K, "global", None, 0.1, 0.4)
    panels = {}
    for dbl in ("1", "0"):
        monkeypatch.setenv("ASB_DOUBLE_PANELS", dbl)
        snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False)
        comp = posComponents(param, snaps)
        comp.deflate_mode = "project"
        comp.compute_components_store_singvalues()
        assert comp.selected_vertices.tolist() == d["idx"].tolist()
        comps, weigs = align_signs(comp.comps, comp.weigs, d["comps"])
        assert relerr(comps, d["comps"]) < 1e-9 and relerr(weigs, d["weigs"]) < 1e-9
        assert relerr(comp.measures_at_largeDeforVerts[:, 1:], d["measures"][:, 1:]) < 1e-8
        panels[dbl] = snaps._engine.deflate_stats()["panels"]
    assert panels["1"] <= panels["0"], panels          # reads of X (equal on data this small, fewer at config-4 size)


def test_project_mode_stepwise_panel_protocol(monkeypatch):
    """The multi-rank panel protocol (asb_panel_* steps, torch exchange buffers, assemble) driven on
    ONE GPU must reproduce the fused single-rank run bit for bit (the fused run without unproven steps, which the
    stepwise protocol does not take: the panel boundaries, and with them the rounding, would differ)."""
    import torch
    monkeypatch.setenv("ASB_SPEC_PANELS", "0")
    rng = np.random.default_rng(33)
    F, N, K = 96, 4000, 20
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    param = _param(vertPos_numComponents=K)
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    outs = []
    for stepwise in (False, True):
        # the stepwise path exchanges torch tensors: run the engine on torch's current (null) stream, as the
        # multi-rank driver does, so kernels and torch copies are stream-ordered
        eng = HipEngine(0, stream=0) if stepwise else None
        snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False, engine=eng)
        comp = posComponents(param, snaps)
        comp.deflate_mode = "project"
        comp._stepwise_panels = stepwise
        comp.compute_components_store_singvalues()
        outs.append((comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(),
                     comp.measures_at_largeDeforVerts.copy()))
    assert outs[0][0].tolist() == outs[1][0].tolist()
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert relerr(outs[1][3], outs[0][3]) < 1e-12


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode,support", [("project", "global"), ("residual", "global"), ("residual", "local")])
def test_multirank_hip_path_on_one_gpu(world, mode, support):
    """`world` vertex shards, each with its own HipEngine, on ONE GPU (threads + emulated collectives,
    tests/thread_comm.py): the real multi-rank HIP kernels == the single-rank run == the oracle."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    if support == "local":
        rest_v, tris = orc.synth_mesh(12, 17, seed=4)      # N = 206
        verts = orc.synth_snapshots(rest_v, 48, rank=6, seed=4, kind="bumps")
        K = 6
    else:
        rng = np.random.default_rng(5)
        verts, tris, K = rng.uniform(-1, 1, size=(64, 5003, 3)), None, 24     # N above the candidate capacity, uneven shards
    param = _param(vertPos_numComponents=K, q_support=support, vertPos_smooth_max_dist=0.35)

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0, stream=0), comm=comm)
            comp = posComponents(param, snaps)
            comp.deflate_mode = mode
            comp.compute_components_store_singvalues()
        return (comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(),
                comp.measures_at_largeDeforVerts.copy(), snaps.pre_scale_factor)

    outs = run_ranks(world, rank_fn)
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris) if support == "local" else None
    ref = orc.extract_k_components(pre["snapTensor"], K, support, geo, 0.1, 0.35)
    for idx, comps, weigs, meas, psf in outs:
        assert abs(psf - pre["pre_scale_factor"]) < 1e-12 * psf
        assert idx.tolist() == ref["idx"].tolist()
        if support == "global":
            comps, weigs = align_signs(comps, weigs, ref["comps"])
        assert relerr(comps, ref["comps"]) < 1e-8
        assert relerr(weigs, ref["weigs"]) < 1e-8
        assert relerr(meas, ref["measures"]) < 1e-7
    for o in outs[1:]:           # every rank ends with the same replicated results
        assert np.array_equal(o[2], outs[0][2]) and np.array_equal(o[1], outs[0][1])


@pytest.mark.parametrize("coop", ["0", "1"])
def test_eight_ranks_projection_protocol_on_one_gpu(coop, monkeypatch):
    """The panel protocol with EIGHT vertex shards (the node size the scaling bench runs): per-rank exports, the global
    threshold selection over 8 x capacity energies, the packed all-gather and its assembly.  Eight contexts share one
    GPU here: once with the panel's inner loop as the two-kernel form, once with the co-resident kernel k_panel_multi ON
    over a reduced grid (256 candidates = 64 blocks per launch; every rank runs the identical steps on the assembled rows
    and none of them may time out)."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    monkeypatch.setenv("ASB_PANEL_COOP", coop)
    if coop == "1":
        monkeypatch.setenv("ASB_M_TARGET", "256")
    rng = np.random.default_rng(58)
    verts, K = rng.uniform(-1, 1, size=(72, 12011, 3)), 30
    param = _param(vertPos_numComponents=K)

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0, stream=0), comm=comm)
            comp = posComponents(param, snaps)
            comp.deflate_mode = "project"
            comp.compute_components_store_singvalues()
        return comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(), snaps._engine.deflate_stats()

    outs = run_ranks(8, rank_fn)
    pre = orc.prepare_snapshots(verts, "first", True)
    ref = orc.extract_k_components(pre["snapTensor"], K, "global", None, 0.1, 0.35)
    assert outs[0][3]["panels"] >= 3
    if coop == "1":
        assert all(o[3]["coop_fallbacks"] == 0 for o in outs)
    for idx, comps, weigs, _ in outs:
        assert idx.tolist() == ref["idx"].tolist()
        comps, weigs = align_signs(comps, weigs, ref["comps"])
        assert relerr(comps, ref["comps"]) < 1e-8 and relerr(weigs, ref["weigs"]) < 1e-8
    for o in outs[1:]:
        assert np.array_equal(o[2], outs[0][2]) and np.array_equal(o[1], outs[0][1])


def test_multirank_splocs_on_one_gpu():
    """SPLOCS over 2 vertex shards on one GPU: partial Gram matrices summed over ranks."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    rest_v, tris = orc.synth_mesh(12, 17, seed=9)
    verts = orc.synth_snapshots(rest_v, 40, rank=6, seed=9, kind="bumps")
    K = 6
    param = _param(vertPos_numComponents=K, q_support="local", vertPos_bases_type="SPLOCS",
                   vertPos_smooth_max_dist=0.4, splocs_max_itrs=3, splocs_admm_num_itrs=4, splocs_lambda=2.0,
                   splocs_rho=10.0)

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0, stream=0), comm=comm)
            comp = posComponents(param, snaps)
            comp.compute_components_store_singvalues()
        return comp.splocs_trace.copy(), comp.splocs_comps.copy(), comp.splocs_centres.copy()

    outs = run_ranks(2, rank_fn)
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris)
    d = orc.extract_k_components(pre["snapTensor"], K, "local", geo, 0.1, 0.4)
    s = orc.splocs_glob_optimization(pre["snapTensor"], d["comps"], d["weigs"], d["R"], geo, 0.1, 0.4, 3, 4, 2.0, 10.0)
    for trace, C, cen in outs:
        assert cen.tolist() == s["idx"].tolist()
        assert np.allclose(trace, s["trace"], rtol=1e-8)
        assert relerr(C, s["C"]) < 1e-8


def _slice_signs(got, ref):
    """orth / qr columns carry LAPACK's arbitrary sign per (component, dimension)."""
    out = got.copy()
    for k in range(got.shape[0]):
        for l in range(3):
            if np.dot(out[k, :, l], ref[k, :, l]) < 0:
                out[k, :, l] *= -1
    return out


def test_orthogonal_post_process_vs_reference_golden():
    """q_standarize + q_orthogonal + q_massWeight against the reference's post-processed basis.
    Global-support components have an arbitrary sign and `+ mean` is applied before `orth`, so the
    reference's own signs are installed first (comps setter -> device upload)."""
    g = load_golden("pca_global_avg_mass_orth")
    param = _param(g)
    assert param.q_orthogonal and param.q_massWeight
    snaps, comp = _run(g["verts"], g["tris"], param, g["mass"])
    comps, _ = align_signs(comp.comps, comp.weigs, g["comps"])
    assert relerr(comps, g["comps"]) < TOL
    comp.comps = comps                                  # caller-assigned basis: uploaded by post_process
    comp.post_process_components()                      # also runs the U^T M U = I assertion (:299)
    got = _slice_signs(comp.comps, g["comps_post"])
    assert relerr(got, g["comps_post"]) < 1e-8
    assert relerr(comp.test_basesSingVals(), g["bases_sing_vals"]) < 1e-8
    M = orc.utmu(comp.comps, g["mass"])
    assert np.allclose(M, np.eye(comp.numComp)[None], atol=1e-9)


def test_orthogonal_vs_oracle_local_support():
    """Local support has canonical signs: deflation -> unscale -> orth on the device vs scipy's orth."""
    rest_v, tris = orc.synth_mesh(14, 20, seed=12)
    verts = orc.synth_snapshots(rest_v, 50, rank=8, seed=12, kind="bumps")
    K = 10
    param = _param(vertPos_numComponents=K, q_support="local", vertPos_smooth_max_dist=0.4, q_orthogonal=True)
    snaps, comp = _run(verts, tris, param)
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris)
    d = orc.extract_k_components(pre["snapTensor"], K, "local", geo, 0.1, 0.4)
    comp.post_process_components()
    ref = orc.post_process_components(d["comps"], pre["pre_scale_factor"], pre["mean"], orthogonal=True)
    got = _slice_signs(comp.comps, ref)
    assert relerr(got, ref) < 1e-7
    for l in range(3):
        assert np.allclose(comp.comps[:, :, l] @ comp.comps[:, :, l].T, np.eye(K), atol=1e-10)


def _cparam(K, orth, tmp, std=True):
    return types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=1,
                                 constProj_massWeight=False, constProj_standarize=std, constProj_orthogonal=orth,
                                 constProj_basis_type="pod_vectorized", deim_desired_num_components=K,
                                 constProj_store_sing_val=True, constProj_output_directory=str(tmp), name="c5",
                                 constProj_name="verts")


def _run_constraints(frames, K, orth, tmp):
    from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
    param = _cparam(K, orth, tmp)
    ns = nonlinearSnapshots(param, frames=frames)
    ns.config()
    ns.snapshots_prepare()
    cc = constraintsComponents(param, ns)
    cc.config()
    cc.compute_components_store_singvalues()
    return ns, cc


@pytest.mark.parametrize("name", ["pod_deim_small", "pod_deim_small_qr"])
def test_pod_deim_vs_reference_golden(name, tmp_path):
    """Config-5 path (standardise -> pod_vectorized -> post-process [-> qr] -> DEIM) against the reference."""
    g = load_golden(name)
    K, orth = int(g["K"]), bool(g["orthogonal"])
    ns, cc = _run_constraints(g["frames"], K, orth, tmp_path)
    assert relerr(ns.snapTensor, g["snapTensor"]) < 1e-12
    S = cc.singular_values
    big = g["S"] > 1e-6 * g["S"][0]           # the Gram route resolves sigma down to ~1e-8 sigma_max
    assert relerr(S[big], g["S"][big]) < 1e-9
    comps = cc.comps.copy()
    for k in range(K):                        # singular vectors: arbitrary sign
        if np.vdot(comps[k], g["comps"][k]) < 0:
            comps[k] *= -1
    assert relerr(comps, g["comps"]) < 1e-8
    lines = open(str(tmp_path / "c5_verts_constrprojBases_pcaExtraction_singValues.csv")).read().splitlines()
    assert lines[0] == "component,singVal" and len(lines) == 1 + g["frames"].shape[0]
    # install the reference's signs, post-process, DEIM
    cc.comps = comps
    cc.post_process_components()
    got = _slice_signs(cc.comps, g["comps_post"]) if orth else cc.comps
    assert relerr(got, g["comps_post"]) < 1e-8
    assert relerr(ns.snapTensor, g["snapTensor_post"]) < 1e-12
    if orth:        # install the reference's column signs before DEIM (the residuals depend on them)
        ns._engine.components_upload(np.ascontiguousarray(got))
        cc._comps, cc._comps_on_device = got, True
    cc.deim()
    assert cc.geom_Pt.tolist() == g["Pt"].tolist()
    assert cc.geom_alpha.tolist() == g["alpha"].tolist()
    assert cc.geom_alpha_ranges.tolist() == g["alpha_ranges"].tolist()
    cc.store_components_n_interpol_points()
    z = np.load(str(tmp_path / "components_interpol_alphas_interpol_verts_interpol_alpha_ranges.npz"))
    assert sorted(z.files) == sorted(["components", "interpol_alphas", "Pt", "interpol_verts", "interpol_alpha_ranges"])


def test_pod_deim_vs_oracle_medium(tmp_path):
    rng = np.random.default_rng(17)
    ep, F, K = 3000, 96, 24
    modes = rng.normal(size=(30, ep, 3))
    coef = rng.normal(size=(F, 30)) * (0.75 ** np.arange(30))[None]
    frames = 0.2 + np.tensordot(coef, modes, (1, 0)) + 1e-6 * rng.normal(size=(F, ep, 3))
    ns, cc = _run_constraints(frames, K, False, tmp_path)
    pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
    pod = orc.pod_vectorized(pre["snapTensor"], K)
    assert relerr(cc.singular_values[:K], pod["S"][:K]) < 1e-10
    comps = cc.comps.copy()
    for k in range(K):
        if np.vdot(comps[k], pod["comps"][k]) < 0:
            comps[k] *= -1
    assert relerr(comps, pod["comps"]) < 1e-7
    cc.comps = comps
    cc.post_process_components()
    post, _ = orc.post_process_constraint_components(pod["comps"], pre["snapTensor"], pre["pre_scale_factor"], pre["mean"])
    assert relerr(cc.comps, post) < 1e-7
    cc.deim()
    dm = orc.deim(post, 1)
    assert cc.geom_Pt.tolist() == dm["Pt"].tolist()


def test_multirank_pod_deim_on_one_gpu(tmp_path):
    """Config-5 path over 2 row shards on one GPU: Gram all-reduce, sharded back-projection, global DEIM arg-max."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, constraintsComponents, nonlinearSnapshots
    from thread_comm import run_ranks
    rng = np.random.default_rng(23)
    ep, F, K = 1501, 64, 26          # K reaches into the 1e-6 noise floor: the Rayleigh-Ritz refinement runs (B all-reduced)
    frames = 0.1 + np.tensordot(rng.normal(size=(F, 20)) * (0.7 ** np.arange(20))[None], rng.normal(size=(20, ep, 3)), (1, 0)) \
        + 1e-6 * rng.normal(size=(F, ep, 3))

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            param = _cparam(K, True, tmp_path)
            param.constProj_store_sing_val = False
            ns = nonlinearSnapshots(param, frames=frames, engine=HipEngine(0, stream=0), comm=comm)
            ns.config()
            ns.snapshots_prepare()
            cc = constraintsComponents(param, ns)
            cc.config()
            cc.compute_components_store_singvalues()
            cc.post_process_components()
            cc.deim()
        return cc.singular_values.copy(), cc.comps.copy(), cc.geom_Pt.copy()

    outs = run_ranks(2, rank_fn)
    pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
    pod = orc.pod_vectorized(pre["snapTensor"], K)
    assert pod["S"][0] > 3e3 * pod["S"][K - 1]
    for S, comps, Pt in outs:
        assert relerr(S[:K], pod["S"][:K]) < 1e-10
        assert np.abs(S[:K] - pod["S"][:K]).max() < 1e-12 * pod["S"][0]
        for l in range(3):       # orthonormal per dimension after CholeskyQR2
            assert np.allclose(comps[:, :, l] @ comps[:, :, l].T, np.eye(K), atol=1e-10)
        assert Pt.tolist() == outs[0][2].tolist() and len(set(Pt.tolist())) == K


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("N,F,K", [(3, 5, 2), (1, 7, 1), (40, 1, 1), (17, 33, 9)])
def test_tiny_and_ragged_shapes(N, F, K, mode):
    """Edge shapes: fewer vertices than a wave, one frame, one vertex, F and N not multiples of anything."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(N * 100 + F)
    X = rng.normal(size=(F, N, 3))
    e = HipEngine(0)
    e.upload(X, 0, N)
    e.deflate_begin(K, False, mode)
    e.run_global(0, K)
    r = e.results()
    e.close()
    ref = orc.extract_k_components(X, K)
    assert r["idx"].tolist() == ref["idx"].tolist()
    comps, weigs = align_signs(r["comps"], r["weigs"], ref["comps"])
    assert relerr(comps, ref["comps"]) < 1e-9 and relerr(weigs, ref["weigs"]) < 1e-9


@pytest.mark.parametrize("mode", [0, 1])
def test_rank_exhausted_does_not_hang(mode):
    """More components than the data's rank (F = 4 frames, K = 8): the residual collapses to rounding noise; the
    reference keeps going on noise (or NaN).  The device path must return (finite or not) or raise -- never hang."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(1)
    X = rng.normal(size=(4, 30, 3))
    e = HipEngine(0)
    e.upload(X, 0, 30)
    e.deflate_begin(8, False, mode)
    try:
        e.run_global(0, 8)
        r = e.results()
        ref = orc.extract_k_components(X, 8)
        assert r["idx"][:4].tolist() == ref["idx"][:4].tolist()      # the meaningful part agrees
    except RuntimeError as ex:
        assert "no progress" in str(ex) or "status -5" in str(ex)
    finally:
        e.close()


def test_multirank_from_device_shards_like_bench():
    """bench.py's multi-rank data path: every rank adopts its OWN device-resident shard (from_device) and the
    projection-mode deflation runs over the shards; result == the single-rank run on the concatenated tensor."""
    import contextlib
    import io
    import torch
    from animsnapbases_amd import HipEngine, partition, posComponents, posSnapshots
    from thread_comm import run_ranks
    F, N, K, world = 48, 6001, 20, 3
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    Xfull = torch.rand((F, N, 3), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    torch.cuda.synchronize()
    param = _param(vertPos_numComponents=K)

    def rank_fn(rank, comm):
        v0, n = partition(N, world)[rank]
        shard = Xfull[:, v0:v0 + n, :].contiguous()
        torch.cuda.synchronize()
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_device(shard.data_ptr(), F, n, rest_shape="first", standarize=True,
                                             engine=HipEngine(0, stream=0), comm=comm, keepalive=shard)
            comp = posComponents(param, snaps)
            comp.compute_components_store_singvalues()
        assert snaps.nVerts == N and snaps._shards == partition(N, world)
        return comp.selected_vertices.copy(), comp.weigs.copy(), comp.measures_at_largeDeforVerts.copy(), comp.comps.copy()

    outs = run_ranks(world, rank_fn)
    with contextlib.redirect_stdout(io.StringIO()):
        one = posSnapshots.from_device(Xfull.data_ptr(), F, N, rest_shape="first", standarize=True)
        c1 = posComponents(param, one)
        c1.compute_components_store_singvalues()
    for idx, weigs, meas, comps in outs:
        assert idx.tolist() == c1.selected_vertices.tolist()
        assert relerr(weigs, c1.weigs) < 1e-10 and relerr(meas, c1.measures_at_largeDeforVerts) < 1e-9
        assert relerr(comps, c1.comps) < 1e-10


def test_device_geodesics_vs_superlu():
    """Heat-method geodesics with batched Jacobi-PCG on the device vs the SuperLU path (reference's solver)."""
    from animsnapbases_amd import GeodesicDistanceComputation, HipEngine
    V, T = orc.synth_mesh(30, 44, seed=2)           # 1322 vertices
    host = GeodesicDistanceComputation(V, T)
    e = HipEngine(0)
    dev = GeodesicDistanceComputation(V, T, engine=e)
    src = [0, 5, 77, 640, 1321] + list(range(100, 170))       # 75 sources: two device batches
    a, b = host.solve_many(src), dev.solve_many(src)
    assert relerr(b, a) < 1e-8
    assert relerr(dev(640), host(640)) < 1e-8
    assert all(it[0] > 0 and it[1] > 0 for it in dev.last_iterations)
    o = orc.Geodesics(V, T)
    assert relerr(dev(77), o(77)) < 1e-8
    e.close()


def test_local_support_with_device_geodesics(monkeypatch):
    monkeypatch.setenv("ASB_GEODESIC", "device")
    g = load_golden("pca_local_small")
    snaps, comp = _run(g["verts"], g["tris"], _param(g))
    assert snaps.compute_geodesic_distance._engine is not None
    print("CG iterations", snaps.compute_geodesic_distance.last_iterations, relerr(comp.comps, g["comps"]))
    assert comp.selected_vertices.tolist() == g["idx"].tolist()
    assert relerr(comp.comps, g["comps"]) < 1e-7 and relerr(comp.weigs, g["weigs"]) < TOL


@pytest.mark.parametrize("n,k", [(1, 1), (2, 2), (3, 2), (5, 5), (64, 7), (65, 65), (130, 20), (777, 33), (1500, 64), (1700, 40), (2601, 64)])
def test_device_symmetric_eigensolver(n, k):
    """asb_sym_tridiag + LAPACK MRRR on T + asb_sym_backtransform against numpy.linalg.eigh: T is orthogonally
    similar to A (same spectrum), and the back-transformed vectors are A's eigenvectors.  n >= 1536: the reflectors come in
    panels of 32 from the co-resident kernel (k_td_panel), the last ~500 from the two-launch loop."""
    import torch
    from scipy.linalg import eigh_tridiagonal
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, max(n // 2, 1)))
    A = B @ B.T + 1e-3 * np.diag(rng.uniform(size=n))          # low rank + small diagonal: clustered small eigenvalues
    A = 0.5 * (A + A.T)
    lam_ref, V_ref = np.linalg.eigh(A)
    e = HipEngine(0, stream=0)
    Ad = torch.from_numpy(A.copy()).cuda()
    d, off = e.sym_tridiag(n, Ad.data_ptr())
    if n == 1:
        assert d[0] == A[0, 0]
        e.close()
        return
    lam = eigh_tridiagonal(d, off, eigvals_only=True)
    assert np.allclose(lam, lam_ref, rtol=0, atol=1e-12 * abs(lam_ref).max())
    if n > 2:
        lamk, Z = eigh_tridiagonal(d, off, select='i', select_range=(n - k, n - 1), lapack_driver='stemr')
        V = e.sym_backtransform(n, Z, Ad.data_ptr())
        assert np.allclose(V.T @ V, np.eye(k), atol=1e-11)
        assert np.abs(A @ V - V * lamk[None]).max() < 1e-11 * abs(lam_ref).max()
    e.close()


def test_tridiagonalisation_panel_exchange_timeout_falls_back(monkeypatch):
    """k_td_panel's grid exchange forced to time out (one block never posts its word): the matrix is restored from its copy and
    the two-launch loop does the whole reduction -- same spectrum, one fall-back counted."""
    import torch
    from scipy.linalg import eigh_tridiagonal
    from animsnapbases_amd import HipEngine
    n = 1700
    rng = np.random.default_rng(3)
    B = rng.normal(size=(n, 300))
    A = B @ B.T + 1e-3 * np.diag(rng.uniform(size=n))
    A = 0.5 * (A + A.T)
    lam_ref = np.linalg.eigvalsh(A)
    monkeypatch.setenv("ASB_TD_TEST_STALL", "1")
    e = HipEngine(0, stream=0)
    Ad = torch.from_numpy(A.copy()).cuda()
    d, off = e.sym_tridiag(n, Ad.data_ptr())
    fb = e.deflate_stats()["coop_fallbacks"]
    e.close()
    assert fb >= 1
    lam = eigh_tridiagonal(d, off, eigvals_only=True)
    assert np.allclose(lam, lam_ref, rtol=0, atol=1e-12 * abs(lam_ref).max())


@pytest.mark.parametrize("ep,F", [(5, 3), (211, 130), (777, 300), (1501, 517)])
def test_pod_gram_mfma_tiles(ep, F):
    """G = A^T A from the LDS-tiled f64-MFMA kernel (128 x 128 tiles above the diagonal, mirrored; split over row
    slabs) against NumPy, at sizes that leave partial tiles, partial 16-row stages and several slabs."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(ep + F)
    frames = rng.normal(size=(F, ep, 3))
    e = HipEngine(0)
    e.upload(frames, 0, ep)
    G = e.pod_gram()
    e.close()
    A = frames.reshape(F, -1)
    ref = A @ A.T
    assert np.array_equal(G, G.T)
    assert np.abs(G - ref).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("n", [1, 7, 16, 100, 128, 129, 300, 1000])
def test_device_spd_inverse(n):
    """Blocked Gauss-Jordan inverse (LDS block inverse + f64-MFMA GEMM sweeps) against numpy.linalg.inv."""
    import ctypes
    from animsnapbases_amd import HipEngine
    from animsnapbases_amd._lib import ptr
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n))
    A = B @ B.T + 0.05 * n * np.eye(n)
    e = HipEngine(0)
    out = np.empty((n, n))
    e._ck(e.lib.asb_test_spd_inverse(e.h, ptr(np.ascontiguousarray(A)), n, ptr(out)))
    e.close()
    ref = np.linalg.inv(A)
    assert np.abs(out - ref).max() < 1e-12 * np.abs(ref).max() * np.linalg.cond(A)
    assert np.abs(out @ A - np.eye(n)).max() < 1e-11


def _run_blocks(frames, K, p, tmp, comm=None, engine=None):
    from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
    param = _cparam(K, False, tmp)
    param.constProj_basis_type = "pca_blocks"
    param.constProj_p_size = p
    ns = nonlinearSnapshots(param, frames=frames, comm=comm, engine=engine)
    ns.config()
    ns.snapshots_prepare()
    cc = constraintsComponents(param, ns)
    cc.config()
    cc.compute_components_store_singvalues()
    return ns, cc


def _check_blocks(cc, ref_comps, ref_weigs, ref_meas, ref_points, ref_blocks, tol=TOL):
    assert cc.largeDeforPoints.tolist() == np.asarray(ref_points).tolist()
    assert cc.largeDeforBlocks.tolist() == np.asarray(ref_blocks).tolist()
    sgn = np.sign(np.sum(cc.weigs * ref_weigs, axis=0))
    assert relerr(cc.weigs * sgn[None], ref_weigs) < tol
    assert relerr(cc.comps * sgn[:, None, None], ref_comps) < tol
    assert relerr(cc.measures_at_largeDeforVerts, ref_meas) < 1e-7


@pytest.mark.parametrize("p", [1, 3])
def test_pca_blocks_vs_reference_golden(p, tmp_path):
    """constProj_basis_type 'pca_blocks' (constraintsComponents.py:324-412) against the unmodified reference; the CSV too."""
    g = load_golden("pca_blocks_p%d" % p)
    ns, cc = _run_blocks(g["frames"], int(g["K"]), p, tmp_path)
    assert cc.numComp == int(g["numComp"])
    _check_blocks(cc, g["comps"], g["weigs"], g["measures"], g["points"], g["blocks"])
    import csv
    rows = list(csv.reader(open(tmp_path / "c5_verts_constrprojBases_pcaExtraction_singValues.csv")))
    assert rows[0] == ['component', 'idx', 'residual_matrix_norm'] + ['singVal%d' % i for i in range(p)]
    assert len(rows) == 1 + int(g["K"])
    got = np.array([[float(x) for x in r] for r in rows[1:]])
    assert relerr(got, g["measures"]) < 1e-7


@pytest.mark.parametrize("p,e,F,K", [(1, 6000, 48, 20), (2, 1500, 33, 9), (4, 700, 26, 5)])
def test_pca_blocks_vs_oracle(p, e, F, K, tmp_path):
    rng = np.random.default_rng(p * 7 + e)
    frames = rng.uniform(-1, 1, size=(F, e * p, 3))
    ns, cc = _run_blocks(frames, K, p, tmp_path)
    pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
    r = orc.pca_blocks(pre["snapTensor"], K, p)
    _check_blocks(cc, r["comps"], r["weigs"], r["measures"], r["points"], r["blocks"], tol=1e-8)


@pytest.mark.parametrize("p", [1, 2])
def test_multirank_pca_blocks_on_one_gpu(p, tmp_path):
    """'pca_blocks' over 2 row shards on one GPU: p = 1 through the panel protocol, p = 2 through block arg-max
    all-gathers and forced-row record exchanges."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine
    from thread_comm import run_ranks
    rng = np.random.default_rng(77 + p)
    e, F, K = 2600, 40, 10
    frames = rng.uniform(-1, 1, size=(F, e * p, 3))

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            ns, cc = _run_blocks(frames, K, p, tmp_path, comm=comm, engine=HipEngine(0, stream=0))
            cc.comps                # gathered over the ranks: has to happen while every rank is still here
        return cc

    outs = run_ranks(2, rank_fn)
    pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
    r = orc.pca_blocks(pre["snapTensor"], K, p)
    for cc in outs:
        _check_blocks(cc, r["comps"], r["weigs"], r["measures"], r["points"], r["blocks"], tol=1e-8)


@pytest.mark.parametrize("N,F", [(1300, 1100), (1300, 96), (900, 1500)])
def test_panel_kernel_candidate_capacity(N, F):
    """The co-resident panel kernel holds one candidate per wave.  With every vertex a candidate (N below the buffer
    capacity) and F > 1024 only 1024 waves are resident: N = 1300 must go through the two-kernel fallback of that
    panel, N = 900 and small F through the register kernel -- all three must give the oracle's sequence."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(N + F)
    K = 12
    X = rng.normal(size=(F, N, 3))
    e = HipEngine(0)
    e.upload(X, 0, N)
    e.deflate_begin(K, False, 1)
    e.run_global(0, K)
    r = e.results()
    e.close()
    ref = orc.extract_k_components(X, K)
    assert r["idx"].tolist() == ref["idx"].tolist()
    comps, weigs = align_signs(r["comps"], r["weigs"], ref["comps"])
    assert relerr(comps, ref["comps"]) < 1e-9 and relerr(weigs, ref["weigs"]) < 1e-9


def test_orthogonal_large_K_vs_scipy_orth():
    """q_orthogonal with K = 160 > 128 (bunny's configuration has 200): the K x K eigen-problem runs on the host, the
    N x K products on the device; result = scipy's SVD-based orth of the same (post-processed) basis, U^T U = I."""
    from scipy.linalg import orth
    from animsnapbases_amd import posComponents, posSnapshots
    rng = np.random.default_rng(160)
    F, N, K = 200, 900, 160
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    param = _param(vertPos_numComponents=K, q_orthogonal=True)
    snaps, comp = _run(verts, None, param)
    pre_orth = comp.comps / snaps.pre_scale_factor + snaps.mean[None]           # what post-processing orthogonalises
    comp.post_process_components()
    for l in range(3):
        ref = orth(pre_orth[:, :, l].T).T
        got = comp.comps[:, :, l]
        sg = np.sign(np.sum(got * ref, axis=1))
        assert relerr(got * sg[:, None], ref) < 1e-7
        assert np.allclose(got @ got.T, np.eye(K), atol=1e-10)


def test_constraints_qr_large_K(tmp_path):
    """constProj_orthogonal with K = 200 (the reference's example configurations use 200 - 1000): CholeskyQR2 with the
    K x K Cholesky on the host == scipy's economic QR up to column signs."""
    import scipy.linalg as sla
    rng = np.random.default_rng(200)
    ep, F, K = 1500, 260, 200
    frames = 0.1 + rng.normal(size=(F, ep, 3))
    ns, cc = _run_constraints(frames, K, True, tmp_path)
    raw = cc.comps.copy() / ns.pre_scale_factor + ns.mean[None]
    cc.post_process_components()
    for l in range(3):
        ref = sla.qr(raw[:, :, l].T, mode="economic")[0].T
        got = cc.comps[:, :, l]
        sg = np.sign(np.sum(got * ref, axis=1))
        assert relerr(got * sg[:, None], ref) < 1e-8
        assert np.allclose(got @ got.T, np.eye(K), atol=1e-11)


def test_splocs_large_K_vs_oracle():
    """SPLOCS with K = 136 > 128: (W^T W + rho I)^-1 through the blocked Gauss-Jordan inverse instead of the one-block
    Cholesky."""
    rest_v, tris = orc.synth_mesh(16, 24, seed=3)          # 386 vertices
    verts = orc.synth_snapshots(rest_v, 150, rank=140, seed=3, kind="bumps", decay=0.985)     # 136 modes above the noise
    K = 136
    param = _param(vertPos_numComponents=K, q_support="local", vertPos_bases_type="SPLOCS", vertPos_smooth_max_dist=0.4,
                   splocs_max_itrs=2, splocs_admm_num_itrs=3, splocs_lambda=2.0, splocs_rho=10.0)
    snaps, comp = _run(verts, tris, param)
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris)
    d = orc.extract_k_components(pre["snapTensor"], K, "local", geo, 0.1, 0.4)
    assert comp.selected_vertices.tolist() == d["idx"].tolist()
    s = orc.splocs_glob_optimization(pre["snapTensor"], d["comps"], d["weigs"], d["R"], geo, 0.1, 0.4, 2, 3, 2.0, 10.0)
    assert comp.splocs_centres.tolist() == s["idx"].tolist()
    assert np.allclose(comp.splocs_trace, s["trace"], rtol=1e-8)
    assert relerr(comp.splocs_comps, s["C"]) < 1e-8


def _fuzz_case(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1023, 1025, 1600, 2500, 4100]))
    F = int(rng.choice([1, 2, 7, 16, 33, 64, 100, 257, 520, 1030]))
    kind = ["uniform", "lowrank", "dupes", "zeros", "tiny", "huge"][seed % 6]
    if kind == "lowrank":
        r = max(1, min(F, 12))
        X = (rng.normal(size=(F, r)) * (0.8 ** np.arange(r))) @ rng.normal(size=(r, N * 3))
        X = X.reshape(F, N, 3) + 1e-6 * rng.normal(size=(F, N, 3))
    else:
        X = rng.uniform(-1, 1, size=(F, N, 3))
    if kind == "dupes" and N > 4:               # exact ties: the first index must win, as np.argmax does
        X[:, N // 2] = X[:, 1]
        X[:, N - 1] = X[:, 1]
    if kind == "zeros" and N > 3:
        X[:, ::3] = 0.0
    if kind == "tiny":
        X *= 1e-120
    if kind == "huge":
        X *= 1e120
    K = int(max(1, min(rng.integers(1, 40), (min(F, 3 * N) + 1) // 2)))
    return X, K, kind


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("seed", list(range(36)))
def test_fuzz_global_deflation(seed, mode):
    """Seeded sweep over awkward shapes (N, F around wave / tile / buffer boundaries) and data (exact duplicates, zero
    vertices, 1e-120 / 1e+120 scaling, low rank): index sequence == oracle, values to 1e-8."""
    from animsnapbases_amd import HipEngine
    X, K, kind = _fuzz_case(seed)
    ref = orc.extract_k_components(X, K)
    if not np.all(np.isfinite(ref["comps"])):
        pytest.skip("the reference itself produces non-finite values here")
    sig = ref["measures"][:, 1]
    e = HipEngine(0)
    e.upload(X, 0, X.shape[1])
    e.deflate_begin(K, False, mode)
    e.run_global(0, K)
    r = e.results()
    e.close()
    # compare up to the first component whose singular value has dropped to rounding noise relative to the first
    good = int(np.argmax(sig < 1e-9 * sig[0])) if np.any(sig < 1e-9 * sig[0]) else K
    assert good >= 1
    assert r["idx"][:good].tolist() == ref["idx"][:good].tolist(), (kind, X.shape, K)
    comps, weigs = align_signs(r["comps"][:good], r["weigs"][:, :good], ref["comps"][:good])
    assert relerr(comps, ref["comps"][:good]) < 1e-8, (kind, X.shape, K)
    assert relerr(weigs, ref["weigs"][:, :good]) < 1e-8, (kind, X.shape, K)


@pytest.mark.parametrize("mode", ["project", "residual"])
@pytest.mark.parametrize("world,N,F,K", [(2, 2, 9, 2), (3, 4, 20, 3), (3, 64, 33, 10), (2, 1000, 100, 20), (3, 1700, 64, 24),
                                         (2, 3001, 257, 30), (3, 2, 12, 2)])
def test_fuzz_multirank_shards(world, N, F, K, mode):
    """Shard shapes at the edges: one vertex per rank, candidate counts around the buffer capacity -- every rank must end
    with the oracle's result; fewer vertices than ranks (an EMPTY shard) is refused with a clear error."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    rng = np.random.default_rng(world * 1000 + N)
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    param = _param(vertPos_numComponents=K)

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0, stream=0), comm=comm)
            comp = posComponents(param, snaps)
            comp.deflate_mode = mode
            comp.compute_components_store_singvalues()
        return comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy()

    if N < world:                     # an empty shard is refused on every rank, with the reason
        with pytest.raises(ValueError, match="cannot be sharded"):
            run_ranks(world, rank_fn)
        return
    outs = run_ranks(world, rank_fn)
    pre = orc.prepare_snapshots(verts, "first", True)
    ref = orc.extract_k_components(pre["snapTensor"], K)
    for idx, comps, weigs in outs:
        assert idx.tolist() == ref["idx"].tolist()
        comps, weigs = align_signs(comps, weigs, ref["comps"])
        assert relerr(comps, ref["comps"]) < 1e-8 and relerr(weigs, ref["weigs"]) < 1e-8


@pytest.mark.parametrize("seed", list(range(10)))
def test_fuzz_local_support(seed):
    """Local support on random small meshes / frame counts: +-project_weight, device geodesic support maps, residual
    updates -- index sequence and values against the oracle (its geodesics: host SuperLU)."""
    rng = np.random.default_rng(1000 + seed)
    rings, segs = int(rng.integers(3, 14)), int(rng.integers(4, 22))
    F = int(rng.choice([6, 17, 40, 65, 130]))
    rest_v, tris = orc.synth_mesh(rings, segs, seed=seed)
    N = rest_v.shape[0]
    K = int(min(rng.integers(2, 9), F // 2, N // 2))
    verts = orc.synth_snapshots(rest_v, F, rank=max(K + 2, 6), seed=seed, kind="bumps", decay=0.9)
    dmax = float(rng.uniform(0.25, 0.6))
    param = _param(vertPos_numComponents=K, q_support="local", vertPos_smooth_max_dist=dmax)
    snaps, comp = _run(verts, tris, param)
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris)
    ref = orc.extract_k_components(pre["snapTensor"], K, "local", geo, 0.1, dmax)
    assert comp.selected_vertices.tolist() == ref["idx"].tolist(), (rings, segs, F, K)
    assert relerr(comp.comps, ref["comps"]) < 1e-7, (rings, segs, F, K)
    assert relerr(comp.weigs, ref["weigs"]) < 1e-7
    assert relerr(comp.measures_at_largeDeforVerts, ref["measures"]) < 1e-7


@pytest.mark.parametrize("ep,F,K,orth", [(2, 5, 2, False), (3, 7, 3, True), (40, 40, 9, True), (129, 33, 12, False),
                                         (700, 130, 64, True), (257, 300, 40, False), (50, 2, 2, True)])
def test_fuzz_pod_deim_shapes(ep, F, K, orth, tmp_path):
    """POD + post-processing + DEIM on awkward shapes (fewer rows than frames, K = F, one row): singular values, spans
    and DEIM points against the oracle."""
    rng = np.random.default_rng(ep * 7 + F)
    r = max(1, min(F, 3 * ep, 2 * K + 3))
    frames = 0.2 + np.tensordot(rng.normal(size=(F, r)) * (0.85 ** np.arange(r))[None], rng.normal(size=(r, ep, 3)), (1, 0)) \
        + 1e-5 * rng.normal(size=(F, ep, 3))
    K = min(K, F, 3 * ep)
    pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
    pod = orc.pod_vectorized(pre["snapTensor"], K)
    if not pod["S"][K - 1] > 1e-7 * pod["S"][0]:      # more components than the numerical rank (F = 2 frames, rest shape
        with pytest.raises(ArithmeticError, match="numerical rank"):      # "first": rank 1): refused, not returned as noise
            _run_constraints(frames, K, orth, tmp_path)
        return
    ns, cc = _run_constraints(frames, K, orth, tmp_path)
    keep = pod["S"][:K] > 1e-7 * pod["S"][0]                      # the Gram route resolves sigma down to ~1e-8 sigma_max
    assert relerr(cc.singular_values[:K][keep], pod["S"][:K][keep]) < 1e-8
    got, want = cc.comps.reshape(K, -1)[keep], pod["comps"].reshape(K, -1)[keep]
    sg = np.sign(np.sum(got * want, axis=1))
    assert relerr(got * sg[:, None], want) < 1e-6


def test_full_size_config4_properties():
    """BASELINE.json config 4 at FULL size (100 000 vertices x 2000 frames, K = 128; 4.8 GB generated on the device):
    size-independent properties of the result, checked with torch on the GPU --
      * the two independent device algorithms (residual tensor kept / projection panels) select the same 128 vertices
        and give the same basis and weights;
      * the first vertex is the arg-max of the initial energies; all selected vertices are distinct;
      * weights mutually orthogonal; c_k = w_k^T X / |w_k|^2 (the identity the reference satisfies to 9e-16);
      * X - sum_k w_k (x) c_k has the recorded Frobenius norm, and the recorded norms never increase."""
    import torch
    from animsnapbases_amd import posComponents, posSnapshots
    F, N, K = 2000, 100000, 128
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2024)
    outs = {}
    for mode in ("project", "residual"):
        gen.manual_seed(2024)
        Xd = torch.rand((F, N, 3), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_device(Xd.data_ptr(), F, N, "first", True, keepalive=Xd)
            comp = posComponents(_param(vertPos_numComponents=K), snaps)
            comp.deflate_mode = mode
            comp.compute_components_store_singvalues()
        outs[mode] = (comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(),
                      comp.measures_at_largeDeforVerts.copy(), snaps.pre_scale_factor, snaps.mean.copy())
        del comp, snaps, Xd
        torch.cuda.empty_cache()
    idx, comps, weigs, meas, psf, mean = outs["project"]
    idx_r, comps_r, weigs_r, meas_r, _, _ = outs["residual"]
    assert idx.tolist() == idx_r.tolist() and len(set(idx.tolist())) == K
    comps_a, weigs_a = align_signs(comps, weigs, comps_r)
    assert relerr(comps_a, comps_r) < 1e-9 and relerr(weigs_a, weigs_r) < 1e-9
    assert relerr(meas[:, 1:], meas_r[:, 1:]) < 1e-9
    assert (np.diff(meas[:, 2]) <= 0).all()
    G = weigs.T @ weigs
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9 * np.diag(G).min()
    # the standardised tensor again, on the device: (X - X[0]) * psf
    gen.manual_seed(2024)
    Xd = torch.rand((F, N, 3), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1
    Xs = ((Xd - Xd[0:1]) * psf).reshape(F, 3 * N)
    del Xd
    assert int(torch.argmax((Xs.reshape(F, N, 3) ** 2).sum(dim=(0, 2))).item()) == int(idx[0])
    Wd = torch.from_numpy(weigs).cuda()                                # (F, K)
    Cd = torch.from_numpy(comps.reshape(K, 3 * N)).cuda()              # (K, 3N)
    proj = (Wd.T @ Xs) / (Wd * Wd).sum(dim=0)[:, None]
    assert float(torch.linalg.norm(proj - Cd) / torch.linalg.norm(Cd)) < 1e-10
    R = Xs - Wd @ Cd
    assert abs(float(torch.linalg.norm(R)) - meas[-1, 2]) < 1e-7 * meas[-1, 2]


def test_full_size_config5_properties(tmp_path):
    """BASELINE.json config 5 at FULL size (50 000 x 3 rows, 4000 frames, K = 256): U orthonormal, its span captures
    exactly the energy the recorded singular values say (|A - U U^T A|_F^2 = sum_{i > K} sigma_i^2), A^T U = V S, and
    DEIM returns K distinct points."""
    import contextlib
    import io
    import torch
    ep, F, K = 50000, 4000, 256
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    r = 300
    coef = torch.randn((F, r), dtype=torch.float64, device="cuda", generator=gen) * (0.97 ** torch.arange(r, device="cuda", dtype=torch.float64))
    modes = torch.randn((r, ep * 3), dtype=torch.float64, device="cuda", generator=gen)
    frames = (0.1 + coef @ modes + 1e-4 * torch.randn((F, ep * 3), dtype=torch.float64, device="cuda", generator=gen))
    frames = frames.reshape(F, ep, 3).cpu().numpy()
    del coef, modes
    torch.cuda.empty_cache()
    with contextlib.redirect_stdout(io.StringIO()):
        ns, cc = _run_constraints(frames, K, False, tmp_path)
        S = cc.singular_values.copy()
        U = torch.from_numpy(cc.comps.reshape(K, -1)).cuda()                      # rows = left singular vectors
        cc.deim()
    assert len(set(cc.geom_Pt.tolist())) == K
    A = torch.from_numpy(((frames - frames[0:1]) * ns.pre_scale_factor).reshape(F, -1)).cuda()      # (F, 3 ep) = A^T
    del frames
    G = U @ U.T
    assert float((G - torch.eye(K, dtype=torch.float64, device="cuda")).abs().max()) < 1e-9
    P = A @ U.T                                                                    # (F, K) = A^T U = V S
    col = torch.linalg.norm(P, dim=0).cpu().numpy()
    assert relerr(col, S[:K]) < 1e-9
    total = float((A * A).sum())
    assert abs(total - float((S ** 2).sum())) < 1e-9 * total
    resid = float(torch.linalg.norm(A - P @ U) ** 2)
    tail = float((S[K:] ** 2).sum())
    assert abs(resid - tail) < 1e-6 * total and resid < 0.01 * total