"""GPU (-m gpu): what round 2 added to the projection path --
  * initial energies carried over from the standardisation sweep (asb_snapshots_scale / k_scale_energy) instead of a
    read of X per call: same sequence and basis as recomputing them (ASB_E0_REUSE=0), zero energy passes reported;
  * the one-thread-per-row correction kernel (k_correct_rows) against the per-vertex one (ASB_CORRECT_ROWS=0);
  * k_panel_coop's record exchange timing out (forced: ASB_COOP_TEST_STALL=1) -> the panel is redone by the two-kernel
    loop, the context stays on it, results unchanged.
All against the NumPy oracle (posComponents.py:67-122), index sequence bit-exact."""
import types

import numpy as np
import pytest

from conftest import align_signs, relerr
from oracle import asb_oracle as orc

pytestmark = pytest.mark.gpu


def _param(K):
    return types.SimpleNamespace(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                                 q_support="global", vertPos_numComponents=K, store_vertPos_PCA_sing_val=False,
                                 vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, vertPos_rest_shape="first",
                                 name="t", vertPos_output_directory=".")


def _run(verts, K, standarize=True, calls=1):
    from animsnapbases_amd import posComponents, posSnapshots
    snaps = posSnapshots.from_arrays(verts, None, "first", standarize=standarize, massWeight=False)
    comp = posComponents(_param(K), snaps)
    comp.deflate_mode = "project"
    stats = []
    for _ in range(calls):
        comp.compute_components_store_singvalues()
        stats.append(snaps._engine.deflate_stats())
    return comp, stats


def _check(comp, d, tol=1e-9):
    assert comp.selected_vertices.tolist() == d["idx"].tolist()
    comps, weigs = align_signs(comp.comps, comp.weigs, d["comps"])
    assert relerr(comps, d["comps"]) < tol and relerr(weigs, d["weigs"]) < tol
    assert relerr(comp.measures_at_largeDeforVerts[:, 1:], d["measures"][:, 1:]) < 1e-7


@pytest.mark.parametrize("kind", ["uniform", "lowrank"])
def test_initial_energies_from_the_standardisation_sweep(kind, monkeypatch):
    rng = np.random.default_rng(5)
    F, N, K = 130, 9000, 40
    if kind == "uniform":
        verts = rng.uniform(-1, 1, size=(F, N, 3))
    else:
        verts = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=12, seed=5)
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    comp, stats = _run(verts, K, calls=2)
    assert [s["energy_passes"] for s in stats] == [0, 0]          # the energies came with asb_snapshots_scale
    _check(comp, d)
    monkeypatch.setenv("ASB_E0_REUSE", "0")
    comp0, stats0 = _run(verts, K, calls=2)
    assert [s["energy_passes"] for s in stats0] == [1, 1]
    _check(comp0, d)
    assert relerr(comp.comps, comp0.comps) < 1e-12 and relerr(comp.weigs, comp0.weigs) < 1e-12
    # not standardised: nothing scaled the tensor, so the first call reads X once and later calls reuse that
    monkeypatch.delenv("ASB_E0_REUSE")
    compn, statsn = _run(verts, K, standarize=False, calls=3)
    assert [s["energy_passes"] for s in statsn] == [1, 0, 0]
    dn = orc.extract_k_components(orc.prepare_snapshots(verts, "first", False)["snapTensor"], K)
    _check(compn, dn)


def test_row_correction_kernel_matches_the_vertex_one(monkeypatch):
    """Many panels (K = 70 on flat random data: the correction against up to 60 earlier components matters) with and
    without unproven steps; ragged vertex count (not a multiple of 64)."""
    rng = np.random.default_rng(9)
    F, N, K = 90, 20011, 70
    verts = rng.uniform(-1, 1, size=(F, N, 3)) + 0.3 * rng.normal(size=(F, 1, 3))
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    outs = {}
    for rows in ("1", "0"):
        for spec in ("1", "0"):
            monkeypatch.setenv("ASB_CORRECT_ROWS", rows)
            monkeypatch.setenv("ASB_SPEC_PANELS", spec)
            comp, _ = _run(verts, K)
            _check(comp, d)
            outs[(rows, spec)] = (comp.comps.copy(), comp.weigs.copy())
    for spec in ("1", "0"):
        assert relerr(outs[("1", spec)][0], outs[("0", spec)][0]) < 1e-12


def test_panel_kernel_timeout_falls_back_to_the_two_kernel_loop(monkeypatch):
    rng = np.random.default_rng(13)
    F, N, K = 100, 12000, 36
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    monkeypatch.setenv("ASB_COOP_TEST_STALL", "1")
    comp, stats = _run(verts, K, calls=2)
    assert stats[0]["coop_fallbacks"] == 1 and stats[1]["coop_fallbacks"] == 1      # once; then the context stays off it
    _check(comp, d)
    monkeypatch.delenv("ASB_COOP_TEST_STALL")
    ref, st = _run(verts, K)
    assert st[0]["coop_fallbacks"] == 0
    assert relerr(comp.comps, ref.comps) < 1e-11


def test_panel_kernel_shares_the_gpu_with_other_contexts():
    """Four vertex shards, each context on its OWN stream, ASB_PANEL_COOP left on: the co-resident panel kernels of
    different contexts can be in flight together, so a launch may find part of the GPU taken.  Whatever happens -- all
    blocks resident, or the record exchange timing out and the panel redone by the two-kernel loop -- every shard must
    deliver the oracle's sequence and identical weights."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    rng = np.random.default_rng(61)
    verts, K = rng.uniform(-1, 1, size=(80, 16000, 3)), 34

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0), comm=comm)
            comp = posComponents(_param(K), snaps)
            comp.deflate_mode = "project"
            comp.compute_components_store_singvalues()
        return comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(), snaps._engine.deflate_stats()

    outs = run_ranks(4, rank_fn)
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    for idx, comps, weigs, st in outs:
        assert idx.tolist() == d["idx"].tolist()
        comps, weigs = align_signs(comps, weigs, d["comps"])
        assert relerr(comps, d["comps"]) < 1e-8 and relerr(weigs, d["weigs"]) < 1e-8
        assert st["coop_fallbacks"] in (0, 1)
        assert st["guessed_panels"] == 1            # rest shape "first" on noise: the first panel's candidates are guessed
    for o in outs[1:]:
        assert np.array_equal(o[2], outs[0][2])


def test_multirank_panel_kernel_timeout_is_redone_in_lock_step(monkeypatch):
    """Three shards through the multi-rank panel protocol; every context's first co-resident launch is made to time out
    (ASB_COOP_TEST_STALL).  The status rides on the panel's min all-reduce, all ranks switch the kernel off together and
    repeat the panel with the two-kernel loop: oracle sequence, identical weights on every rank, one fallback each."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    monkeypatch.setenv("ASB_COOP_TEST_STALL", "1")
    rng = np.random.default_rng(67)
    verts, K = rng.uniform(-1, 1, size=(64, 9000, 3)), 26

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0, stream=0), comm=comm)
            comp = posComponents(_param(K), snaps)
            comp.deflate_mode = "project"
            comp.compute_components_store_singvalues()
        return comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(), snaps._engine.deflate_stats()

    outs = run_ranks(3, rank_fn)
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    for idx, comps, weigs, st in outs:
        assert idx.tolist() == d["idx"].tolist()
        comps, weigs = align_signs(comps, weigs, d["comps"])
        assert relerr(comps, d["comps"]) < 1e-8 and relerr(weigs, d["weigs"]) < 1e-8
        assert st["coop_fallbacks"] == 1
    for o in outs[1:]:
        assert np.array_equal(o[2], outs[0][2])


@pytest.mark.parametrize("shape", [(130, 9000, 40), (400, 30000, 48)])
def test_first_panel_guessed_from_the_energies_without_the_constant_direction(shape, monkeypatch):
    """Rest shape "first" on noise: half of |X|^2 is each row's own offset, the first components remove it from every
    vertex, and a first panel chosen by the initial energies alone ends after a few steps.  With the guess
    (ASB_FIRST_PANEL_MEAN, default on) the first panel's candidates also come from the energies without the constant
    direction; everything beyond the provable steps is checked by the pass, so sequence and basis are the oracle's either
    way -- and fewer reads of X are needed."""
    F, N, K = shape
    rng = np.random.default_rng(21)
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    comp, st = _run(verts, K, calls=2)
    _check(comp, d)
    assert [s["guessed_panels"] for s in st] == [1, 1]
    monkeypatch.setenv("ASB_FIRST_PANEL_MEAN", "0")
    comp0, st0 = _run(verts, K)
    _check(comp0, d)
    assert st0[0]["guessed_panels"] == 0
    assert relerr(comp.comps, comp0.comps) < 1e-12 and relerr(comp.weigs, comp0.weigs) < 1e-12
    assert st[0]["panels"] <= st0[0]["panels"]
    # rest shape "average" removes the constant direction beforehand: nothing to guess
    monkeypatch.delenv("ASB_FIRST_PANEL_MEAN")
    from animsnapbases_amd import posComponents, posSnapshots
    snaps = posSnapshots.from_arrays(verts, None, "average", standarize=True, massWeight=False)
    p = _param(K)
    p.vertPos_rest_shape = "average"
    compa = posComponents(p, snaps)
    compa.deflate_mode = "project"
    compa.compute_components_store_singvalues()
    assert snaps._engine.deflate_stats()["guessed_panels"] == 0
    da = orc.extract_k_components(orc.prepare_snapshots(verts, "average", True)["snapTensor"], K)
    _check(compa, da)


def test_multirank_first_panel_guess_is_a_collective_decision(monkeypatch):
    """Two shards through the multi-rank protocol with and without the guessed first panel: the decision is taken on
    all-reduced sums, each rank contributes its share of the guessed candidates, the pass checks them on every shard --
    same oracle sequence, same weights on both ranks, no more panels than without."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    rng = np.random.default_rng(71)
    verts, K = rng.uniform(-1, 1, size=(200, 24000, 3)), 40
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0, stream=0), comm=comm)
            comp = posComponents(_param(K), snaps)
            comp.deflate_mode = "project"
            comp.compute_components_store_singvalues()
        return comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(), snaps._engine.deflate_stats()

    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("ASB_FIRST_PANEL_MEAN", on)
        outs = run_ranks(2, rank_fn)
        for idx, comps, weigs, st in outs:
            assert idx.tolist() == d["idx"].tolist()
            comps, weigs = align_signs(comps, weigs, d["comps"])
            assert relerr(comps, d["comps"]) < 1e-8 and relerr(weigs, d["weigs"]) < 1e-8
            assert st["guessed_panels"] == int(on)
        assert np.array_equal(outs[0][2], outs[1][2])
        res[on] = outs[0][3]["panels"]
    assert res["1"] <= res["0"]


@pytest.mark.parametrize("kind", ["uniform", "lowrank_noise"])
def test_reads_with_several_sub_panels_project_on_orthogonalised_weights(kind, monkeypatch):
    """Several sub-panels per read of X (default): the pass projects on weights made orthogonal to every earlier weight
    vector beforehand (k_orth_wt) instead of correcting the coefficients against all earlier columns afterwards
    (ASB_PRE_ORTH=0) -- a second-order difference.  Both against the oracle; the low-rank + noise case is the one where
    the leakage matters at all (components 1e4 apart in strength inside one panel)."""
    rng = np.random.default_rng(31)
    F, N, K = 160, 24000, 60
    if kind == "uniform":
        verts = rng.uniform(-1, 1, size=(F, N, 3))
    else:
        verts = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=20, noise=1e-4, seed=31)
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    outs = {}
    for pre in ("1", "0"):
        monkeypatch.setenv("ASB_PRE_ORTH", pre)
        comp, st = _run(verts, K)
        _check(comp, d, tol=1e-9 if kind == "uniform" else 1e-8)
        outs[pre] = (comp.comps.copy(), comp.weigs.copy(), st[0]["panels"])
    assert relerr(outs["1"][0], outs["0"][0]) < 1e-10 and relerr(outs["1"][1], outs["0"][1]) < 1e-12
    assert outs["1"][2] == outs["0"][2]               # the same panels either way
    if kind == "uniform":
        assert outs["1"][2] <= 4                      # reads of X: 60 components would take four 16-column panels at the least


@pytest.mark.parametrize("kind", ["uniform", "lowrank_noise"])
def test_reads_finished_on_the_device_match_the_step_by_step_path(kind, monkeypatch):
    """By default all sub-panels of a read run in ONE launch (k_panel_multi) and its tiles are enqueued back to back and read
    once (ASB_TILE_CHAIN); with ASB_SUB_CHAIN=0 / ASB_TILE_CHAIN=0 every sub-panel is its own launch (same kernel, rows written
    back in between) and every tile costs a host read.  Same panels, same basis --
    on data where every tile stands (uniform) and on data where reads are cut short by rejections (low rank + noise: the
    tile that does not stand in full is committed the slow way from untouched energies)."""
    rng = np.random.default_rng(41)
    F, N, K = (1100, 9000, 70) if kind == "uniform" else (300, 16000, 70)      # both register layouts of the panel kernel
    if kind == "uniform":
        verts = rng.uniform(-1, 1, size=(F, N, 3))
    else:
        verts = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=24, noise=1e-4, seed=41)
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    outs = {}
    for chain in ("1", "0"):
        monkeypatch.setenv("ASB_SUB_CHAIN", chain)
        monkeypatch.setenv("ASB_TILE_CHAIN", chain)
        comp, st = _run(verts, K)
        _check(comp, d, tol=1e-9 if kind == "uniform" else 1e-8)
        outs[chain] = (comp.comps.copy(), comp.weigs.copy(), comp.measures_at_largeDeforVerts.copy(), st[0])
    # The same components from the same reads.  Bit for bit where both paths run the same kernels on the same tile counts (every
    # tile stands); where reads are cut short the chained path has its pass enqueued on the EXPECTED counts (the four-tile kernel)
    # and the step-by-step path on the counts reached (possibly the three- or two-tile kernel: another summation order): rounding.
    rc, rw = relerr(outs["1"][0], outs["0"][0]), relerr(outs["1"][1], outs["0"][1])
    print("chained against step by step:", rc, rw)
    if kind == "uniform":
        assert np.array_equal(outs["1"][0], outs["0"][0]) and np.array_equal(outs["1"][1], outs["0"][1])
    assert rc < 1e-13 and rw < 1e-13
    assert relerr(outs["1"][2][:, 1:], outs["0"][2][:, 1:]) < 1e-12          # (column sums: other block partials)
    assert outs["1"][3]["panels"] == outs["0"][3]["panels"]
    if kind == "uniform":
        assert outs["1"][3]["panels"] <= 3
