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


def _cparam(K, tmp):
    return types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=1, constProj_massWeight=False,
                                 constProj_standarize=True, constProj_orthogonal=False, constProj_basis_type="pod_vectorized",
                                 deim_desired_num_components=K, constProj_store_sing_val=False, constProj_output_directory=str(tmp),
                                 name="t", constProj_name="v")


@pytest.mark.parametrize("ep,F,r,K,floor", [(400, 120, 20, 44, 1e-9), (1200, 260, 30, 100, 1e-10), (300, 64, 6, 30, 1e-12)])
def test_pod_reaches_into_the_noise_floor_like_svd(ep, F, r, K, floor, tmp_path):
    """constraintsComponents.py:307-316 returns K left vectors whatever the spectrum.  K cuts deep into a noise floor 1e-9 and
    less below the signal -- invisible in the Gram matrix of A (round 3 raised ArithmeticError): the POD goes on level by level
    on the deflated snapshots (asb_pod_deflate_begin).  Against SciPy's svd of A (the oracle): all singular values, the signal
    vectors one by one, the floor vectors one by one where the floor's own gaps determine them and as a subspace always."""
    from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
    rng = np.random.default_rng(ep + F)
    coef = rng.normal(size=(F, r)) * np.logspace(0, -3, r)[None]
    frames = 0.3 + np.tensordot(coef, rng.normal(size=(r, ep, 3)), (1, 0)) + floor * rng.normal(size=(F, ep, 3))
    pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
    pod = orc.pod_vectorized(pre["snapTensor"], K)
    S_ref = pod["S"]
    n_sig = int(np.sum(S_ref > 1e-6 * S_ref[0]))
    assert n_sig < K - 8 and S_ref[K - 1] < 3e-8 * S_ref[0]           # K is far inside the floor, below the Gram route's reach
    param = _cparam(K, tmp_path)
    ns = nonlinearSnapshots(param, frames=frames)
    ns.config()
    ns.snapshots_prepare()
    cc = constraintsComponents(param, ns)
    cc.config()
    cc.compute_components_store_singvalues()
    assert cc.pod_levels >= 2
    S = cc.singular_values
    assert S.shape == S_ref.shape
    assert relerr(S[:n_sig], S_ref[:n_sig]) < 1e-9
    nz = S_ref > 1e-3 * S_ref[n_sig]                                    # (rest shape "first": one exact zero at the end)
    assert relerr(S[nz], S_ref[nz]) < 1e-6, relerr(S[nz], S_ref[nz])
    got, want = cc.comps.reshape(K, -1), pod["comps"].reshape(K, -1)
    assert np.abs(got @ got.T - np.eye(K)).max() < 1e-10
    sg = np.sign(np.sum(got * want, axis=1))
    per = np.linalg.norm(got * sg[:, None] - want, axis=1)
    print("levels %d; per-vector error: signal max %.2e, floor max %.2e (median %.2e)" %
          (cc.pod_levels, per[:n_sig].max(), per[n_sig:].max(), np.median(per[n_sig:])))
    assert per[:n_sig].max() < 1e-7
    # the floor: the span of the first K vectors is determined by the gap between sigma_K and sigma_K+1 alone -- and only as
    # well as a backward-stable SVD determines it: a perturbation eps |A| moves it by eps sigma_0 / (sigma_K - sigma_K+1), which
    # is what two correct algorithms (LAPACK's on A, this one) may differ by (here ~1e-6: sigma_K ~ 3e-8 sigma_0, gap ~ 3e-3)
    gap = (S_ref[K - 1] - S_ref[K]) / S_ref[K - 1]
    cond = 2.2e-16 * S_ref[0] / (S_ref[K - 1] * gap)
    Pg, Pw = got.T @ got, want.T @ want
    print("relative gap at the cut %.2e, conditioning eps sigma_0 / (sigma_K gap) = %.2e, |P - P_ref| = %.2e" %
          (gap, cond, np.linalg.norm(Pg - Pw)))
    if 100.0 * cond < 1e-2:          # (a floor at 1e-12 sigma_0 leaves the cut undetermined for LAPACK as well: cond > 1)
        assert np.linalg.norm(Pg - Pw) < max(1e-7, 100.0 * cond)
    # and every returned floor vector is a singular vector of A to the floor's own accuracy: |A^T u| = its singular value
    A = pre["snapTensor"].reshape(F, -1).T
    rq = np.linalg.norm(A.T @ got.T, axis=0)
    if cond < 1e-2:
        assert relerr(rq[n_sig:], S_ref[n_sig:K]) < 1e-4
    else:                            # floor at rounding level of A: the values are floor-sized, no more can be said
        assert np.all(rq[n_sig:] < 4.0 * S_ref[n_sig]) and np.all(rq[n_sig:] > 0.25 * S_ref[K - 1])


@pytest.mark.parametrize("world,kind", [(3, "lowrank"), (2, "bumps"), (3, "rankdef")])
def test_multirank_read_in_one_exchange_on_structured_data(world, kind):
    """The multi-rank read of round 4 (asb_panel_read_run / _commit: one launch of the panel kernel, one min-all-reduce per read)
    where the shards DISAGREE: on structured data a tile stands on one shard and falls on another, so the local chain of a
    shard runs ahead of the verdict and is rolled back; on rank-deficient data the stall rule fires over the ranks and the run
    continues in the residual protocol.  `world` shards with their own contexts on one GPU (emulated collectives)."""
    import contextlib
    import io
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    from thread_comm import run_ranks
    rng = np.random.default_rng(world * 7 + len(kind))
    N, F = 12001, 96
    if kind == "lowrank":
        verts, K = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=20, noise=1e-4, decay=0.9, seed=3), 40
    elif kind == "bumps":
        verts, K = orc.synth_snapshots(rng.uniform(size=(N, 3)), F, rank=24, noise=1e-4, decay=0.9, seed=4, kind="bumps"), 40
    else:
        verts, K = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=6, noise=1e-13, decay=0.8, seed=5), 30
    param = _param(K)

    def rank_fn(rank, comm):
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False,
                                             engine=HipEngine(0, stream=0), comm=comm)
            comp = posComponents(param, snaps)
            comp.deflate_mode = "project"
            comp.compute_components_store_singvalues()
        return (comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(), comp.measures_at_largeDeforVerts.copy(),
                snaps._engine.deflate_stats())

    outs = run_ranks(world, rank_fn)
    pre = orc.prepare_snapshots(verts, "first", True)
    ref = orc.extract_k_components(pre["snapTensor"], K)
    sig = ref["measures"][:, 1]
    good = int(np.argmax(sig < 1e-7 * sig[0])) if np.any(sig < 1e-7 * sig[0]) else K
    print(kind, "world", world, "determined components", good, "of", K, "| residual switch at", [o[4]["residual_switch_at"] for o in outs],
          "| reads", [o[4]["panels"] + o[4]["refreshes"] for o in outs])
    for idx, comps, weigs, meas, st in outs:
        assert idx[:good].tolist() == ref["idx"][:good].tolist()
        c, w = align_signs(comps[:good], weigs[:, :good], ref["comps"][:good])
        assert relerr(c, ref["comps"][:good]) < 1e-7 and relerr(w, ref["weigs"][:, :good]) < 1e-7
        assert relerr(meas[:good, 1], sig[:good]) < 1e-7
        rec = np.tensordot(weigs, comps, axes=([1], [0]))
        if kind == "rankdef":
            assert st["residual_switch_at"] >= good - 1
            assert relerr(rec, pre["snapTensor"]) < 1e-9
    for o in outs[1:]:           # every rank ends with the same replicated results
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[2], outs[0][2]) and np.array_equal(o[1], outs[0][1])
