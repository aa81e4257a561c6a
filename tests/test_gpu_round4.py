"""GPU (-m gpu), round 4:
  * the streamed download of the basis (HipEngine.components_stream / components_pinned): the pinned basis equals the plain
    download bit for bit on random and on low-rank data (rejections, predicted reads), over two runs on one engine with
    different K; a view handed out earlier is neither freed nor overwritten by a later run, by switching the stream off or
    by closing the engine (the buffer belongs to the views, include/asb.h: asb_components_stream_into);
  * asb_deflate_reserve: a residual-mode run that grows its basis in mid-run equals the run that reserved everything;
  * the stall rule (asb_project_switch_residual): K far beyond the numerical rank -- the run leaves the projection mode, the
    components the data determine are the oracle's (posComponents.py:67-122), the rest reconstruct X to rounding level.
"""
import gc
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


def _data(kind, N, F, seed):
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.uniform(-1, 1, size=(F, N, 3))
    return orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=25, noise=1e-4, decay=0.9, seed=seed)


@pytest.mark.parametrize("kind", ["random", "lowrank"])
def test_streamed_basis_equals_download(kind):
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    N, F = 9000, 96
    eng = HipEngine(0)
    eng.components_stream(True)
    kept = []
    for run, K in enumerate((40, 24, 56)):
        verts = _data(kind, N, F, 100 + run)
        snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False, engine=eng)
        comp = posComponents(_param(K), snaps)
        comp.deflate_mode = "project"
        comp.compute_components_store_singvalues()
        st = eng.deflate_stats()
        pinned = comp.comps                                   # ndarray over the pinned buffer
        plain = eng.results_comps()                           # a fresh device -> host copy
        assert pinned.shape == (K, N, 3)
        assert np.array_equal(pinned, plain), "streamed basis differs from the download (reads %d)" % (st["panels"] + st["refreshes"])
        kept.append((pinned, plain))
        # every earlier view still holds ITS run's basis: a later run took a fresh buffer instead of overwriting it
        for old_view, old_copy in kept[:-1]:
            assert np.array_equal(old_view, old_copy)
    # a run whose views are gone may reuse the buffer; the views that are alive survive stream-off and close
    view, copy = kept[-1]
    sl = view[3:5]                                            # a derived view alone must keep the memory alive, too
    del kept, view, pinned
    gc.collect()
    eng.components_stream(False)
    assert np.array_equal(sl, copy[3:5])
    eng.close()
    gc.collect()
    assert np.array_equal(sl, copy[3:5])


def test_streamed_basis_without_a_stream_raises():
    from animsnapbases_amd import HipEngine
    eng = HipEngine(0)
    with pytest.raises(RuntimeError):
        eng.components_pinned()
    eng.close()


def test_deflate_reserve_keeps_what_the_run_produced():
    from animsnapbases_amd import HipEngine, _lib
    rng = np.random.default_rng(5)
    N, F, K = 700, 40, 12
    X = rng.normal(size=(F, N, 3))

    def run(reserve_at):
        eng = HipEngine(0)
        eng.upload(X, 0, N)
        eng.deflate_begin(K if reserve_at is None else reserve_at, False, _lib.DEFLATE_RESIDUAL)
        for k in range(K):
            if reserve_at is not None and k == eng.K:
                eng.deflate_reserve(min(K, 2 * eng.K))
            eng.pick(k)
            eng.apply(k)
        assert eng.K == K
        res = eng.results(want_comps=True, want_weigs=True)
        eng.close()
        return res

    a, b = run(None), run(3)
    for key in ("comps", "weigs", "idx", "sigma", "normR2_local"):
        assert np.array_equal(a[key], b[key]), key
    # and against the oracle
    d = orc.extract_k_components(X.copy(), K)
    assert a["idx"].tolist() == d["idx"].tolist()


@pytest.mark.parametrize("N,F,rank,K", [(40000, 64, 8, 40), (6000, 48, 5, 30)])
def test_rank_deficient_run_leaves_the_projection_mode(N, F, rank, K, monkeypatch):
    from animsnapbases_amd import posComponents, posSnapshots
    rng = np.random.default_rng(N)
    verts = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=rank, noise=1e-13, decay=0.8, seed=N)
    pre = orc.prepare_snapshots(verts, "first", True)
    d = orc.extract_k_components(pre["snapTensor"], K)
    sig = d["measures"][:, 1]
    good = int(np.argmax(sig < 1e-7 * sig[0])) if np.any(sig < 1e-7 * sig[0]) else K
    assert 3 <= good < K - 10            # K reaches far beyond what the data determine

    def run(fallback):
        monkeypatch.setenv("ASB_STALL_FALLBACK", "1" if fallback else "0")
        snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False)
        comp = posComponents(_param(K), snaps)
        comp.deflate_mode = "project"
        comp.compute_components_store_singvalues()
        return comp, snaps._engine.deflate_stats(), snaps

    comp, st, snaps = run(True)
    reads = st["panels"] + st["refreshes"]
    print("rank %d, K = %d: switched at component %d after %d reads of X" % (rank, K, st["residual_switch_at"], reads))
    assert st["residual_switch_at"] >= good - 1, st
    assert comp.selected_vertices[:good].tolist() == d["idx"][:good].tolist()
    comps, weigs = align_signs(comp.comps[:good], comp.weigs[:, :good], d["comps"][:good])
    assert relerr(comps, d["comps"][:good]) < 1e-7 and relerr(weigs, d["weigs"][:, :good]) < 1e-7
    assert np.allclose(comp.measures_at_largeDeforVerts[:good, 1], sig[:good], rtol=1e-7)
    # all K components together reconstruct the prepared tensor to rounding level, and |R_k| says so
    rec = np.tensordot(comp.weigs, comp.comps, axes=([1], [0]))
    assert relerr(rec, pre["snapTensor"]) < 1e-9
    nr = comp.measures_at_largeDeforVerts[:, 2]
    assert np.all(np.isfinite(nr)) and nr[-1] < 1e-9 * np.linalg.norm(pre["snapTensor"])
    assert np.all(np.diff(nr[:good]) <= 0)
    # the reads the rule saved: without it the panels grind on
    comp0, st0, _ = run(False)
    reads0 = st0["panels"] + st0["refreshes"]
    print("without the rule: %d reads of X" % reads0)
    assert st0["residual_switch_at"] == -1 and reads0 > reads
    assert comp0.selected_vertices[:good].tolist() == d["idx"][:good].tolist()


@pytest.mark.parametrize("rest_shape", ["first", "average"])
def test_device_resident_constraint_frames_with_mass_weights(rest_shape):
    """nonlinear_snapshots.py:74-96 with constProj_massWeight on frames that already sit in HBM (frames_device=): the same
    prepared tensor, mean and scale as the host-array path and as the oracle."""
    import torch
    from animsnapbases_amd import nonlinearSnapshots
    rng = np.random.default_rng(9)
    F, ep = 24, 333
    frames = 0.2 + rng.normal(size=(F, ep, 3)) * rng.uniform(0.2, 1.0, size=(1, ep, 1))
    mass = rng.uniform(0.5, 2.0, size=ep)
    param = types.SimpleNamespace(constProj_rest_shape=rest_shape, constProj_numFrames=F, constProj_p_size=1, constProj_massWeight=True,
                                  constProj_standarize=True, constProj_orthogonal=False, constProj_output_directory=".", name="t",
                                  constProj_name="v")
    host = nonlinearSnapshots(param, frames=frames, mass=mass)
    host.config()
    host.snapshots_prepare()
    Xd = torch.from_numpy(frames).to("cuda:0").contiguous()
    torch.cuda.synchronize()
    devs = nonlinearSnapshots(param, frames_device=(Xd.data_ptr(), F, ep), keepalive=Xd, mass=mass)
    devs.config()
    devs.snapshots_prepare()
    assert np.allclose(devs.massL, np.sqrt(mass))
    assert abs(devs.pre_scale_factor / host.pre_scale_factor - 1.0) < 1e-13
    assert relerr(devs.mean, host.mean) < 1e-14
    assert relerr(devs.snapTensor, host.snapTensor) < 1e-13
    # and the reference's arithmetic: weight, remove the rest shape, scale by 1 / std
    W = frames * np.sqrt(mass)[None, :, None]
    mean = W[0] if rest_shape == "first" else W.mean(axis=0)
    R = W - mean[None]
    assert relerr(devs.snapTensor, R / R.std()) < 1e-12
