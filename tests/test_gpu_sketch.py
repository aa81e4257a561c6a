"""GPU (-m gpu): the sketch predictor of the panel algorithm (animsnapbases_amd/csrc/asb_sketch.hip).

On structured data a read of X commits few of its steps (the ranking reshuffles under its candidates); the columns of the
rejected steps are a sketch of every vertex's residual, and a greedy replay in that space names the next read's candidates.
  * the replay kernel against a NumPy model of it (same arithmetic in f64): predicted winners, scores;
  * its exchange timing out (forced) -> scores fall back to the energies, nothing hangs;
  * end to end on low rank + noise: the selected sequence and the basis are the oracle's (posComponents.py:67-122) with the
    predictor on and off, and the predictor cuts the reads of X.
"""
import ctypes
import types

import numpy as np
import pytest

from conftest import align_signs, relerr
from oracle import asb_oracle as orc

pytestmark = pytest.mark.gpu


def replay_model(Z, E, steps):
    """Z (r, 3n) coordinates on unit directions, E (n) exact energies: the replay of asb_sketch.hip in f64."""
    Z = Z.copy()
    n = E.shape[0]
    tail = np.maximum(E - (Z * Z).sum(0).reshape(n, 3).sum(1), 0.0)
    e = (Z * Z).sum(0).reshape(n, 3).sum(1) + tail
    score = np.zeros(n)
    pred = []
    for _ in range(steps):
        v = int(np.argmax(e))
        if not e[v] > 0:
            break
        score = np.maximum(score, e / e[v])
        pred.append(v)
        A = Z[:, 3 * v:3 * v + 3].copy()
        lam, U = np.linalg.eigh(A.T @ A + tail[v] / 3 * np.eye(3))
        u, lam = U[:, -1], lam[-1]
        q = A @ u
        Z -= np.outer(q, (q @ Z) / lam)
        Z[:, 3 * v:3 * v + 3] = A - np.outer(q, u)
        tail[v] *= 2.0 / 3.0
        e = (Z * Z).sum(0).reshape(n, 3).sum(1) + tail
    return score, pred


def replay_model_f32(Z, E, steps):
    """The same replay in the KERNEL's arithmetic: the sketch, its Gram sums, the deflation and the re-summed energies in
    float32 (the kernel holds the sketch as packed f32 and re-sums a vertex's energy from it every step), the tail E - |Z|^2 from
    the f64 energy once, the winner's 3 x 3 eigen-pair in f64 from f32 Gram sums.  NumPy has no fused multiply-add, so single
    products differ from the kernel's by an ulp: scores agree to ~1e-5 instead of bit for bit."""
    f32 = np.float32
    n = E.shape[0]
    Z = Z.astype(f32).reshape(Z.shape[0], n, 3).copy()          # (r, n, 3)
    own = (Z * Z).sum(axis=(0, 2), dtype=f32)
    tail = np.maximum(E - (Z.astype(np.float64) ** 2).sum(axis=(0, 2)), 0.0).astype(f32)      # (the kernel sums |Z_v|^2 in f64 for this)
    e = own + tail
    score = np.zeros(n, dtype=f32)
    pred = []
    for _ in range(steps):
        v = int(np.argmax(e))
        if not e[v] > 0:
            break
        score = np.maximum(score, e / e[v])
        pred.append(v)
        A = Z[:, v, :].copy()                                    # (r, 3) f32
        G = (A[:, :, None] * A[:, None, :]).sum(axis=0, dtype=np.float64).astype(f32).astype(np.float64)
        lam, U = np.linalg.eigh(G + (np.float64(tail[v]) / 3.0) * np.eye(3))
        u, lam = U[:, -1], lam[-1]
        u = u * (1.0 if u[np.argmax(np.abs(u))] > 0 else -1.0)
        uf = u.astype(f32)
        q = (A * uf[None, :]).sum(axis=1, dtype=f32)             # (r,)
        il = f32(1.0 / lam)
        a = np.einsum("i,ind->nd", q, Z).astype(f32) * il        # (n, 3)
        a[v] = uf
        Z -= q[:, None, None] * a[None, :, :]
        tail[v] *= f32(2.0 / 3.0)
        e = (Z * Z).sum(axis=(0, 2), dtype=f32) + tail
    return score.astype(np.float64), pred


def _predict(eng, cols, wn2, E, steps):
    r, n3 = cols.shape
    n = n3 // 3
    scores = np.zeros(n)
    pred = (ctypes.c_int64 * steps)()
    status = ctypes.c_int(0)
    cols = np.ascontiguousarray(cols)
    wn2 = np.ascontiguousarray(wn2)
    E = np.ascontiguousarray(E)
    eng._ck(eng.lib.asb_test_sketch_predict(eng.h, cols.ctypes.data, wn2.ctypes.data, E.ctypes.data, n, r, steps,
                                            scores.ctypes.data, pred, ctypes.byref(status)))
    return scores, [int(p) for p in pred], status.value


@pytest.mark.parametrize("n,r,steps,tail_rel,tol", [(5000, 40, 30, 1e-6, 5e-4), (70001, 64, 64, 1e-3, 5e-4), (1300, 7, 12, 0.5, 5e-4)])
def test_replay_kernel_against_its_numpy_model(n, r, steps, tail_rel, tol):
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(n)
    # a decaying shared structure: mode j has strength 0.85^j; every vertex takes part in every mode
    Z = rng.normal(size=(r, 3 * n)) * (0.85 ** np.arange(r))[:, None]
    wn2 = rng.uniform(0.5, 2.0, size=r)                  # the kernel is handed coefficients c = z / |w| and |w|^2
    cols = Z / np.sqrt(wn2)[:, None]
    own = (Z * Z).sum(0).reshape(n, 3).sum(1)
    E = own + tail_rel * own.mean() * rng.uniform(0.5, 1.5, size=n)
    eng = HipEngine(0)
    scores, pred, status = _predict(eng, cols, wn2, E, steps)
    assert status == 1
    # Round 3 held the kernel to the f64 model at 5e-3 and to the first HALF of its winners: the tail E - |Z|^2 was a difference
    # of f32 sums, 10 % wrong for a tail of 1e-6.  With |Z_v|^2 summed in f64 for the tail (round 4) the kernel follows the f64
    # model to 5e-4 and names (nearly) all its winners.  A NumPy model in f32 is printed beside it: it is the LESS faithful one --
    # NumPy's f32 sums (pairwise, unfused) are not the kernel's (sequential packed FMAs) -- which is why it is not the bound.
    ms, mp = replay_model(Z, E, steps)
    ms32, mp32 = replay_model_f32(Z, E, steps)
    agree = next((t for t in range(len(mp)) if pred[t] != mp[t]), len(mp))
    print("f64 model: largest score difference %.2e, winners agree for %d of %d steps; NumPy f32 model: %.2e, %d of %d" %
          (np.abs(scores - ms).max(), agree, len(mp), np.abs(scores - ms32).max(),
           next((t for t in range(len(mp32)) if pred[t] != mp32[t]), len(mp32)), len(mp32)))
    assert agree >= (7 * len(mp)) // 8, (agree, len(mp))
    assert np.abs(scores - ms).max() < tol
    eng.close()


def test_replay_on_the_largest_energies_above_one_launch():
    """A shard above the co-resident capacity (256 blocks x 512 vertices): the replay runs on the ~118 000 largest energies --
    a vertex's energy never grows, so who can win soon is among them -- and everyone else keeps score 0; the subset is an upper
    set of the energies, and on it the kernel agrees with the model run on the subset alone."""
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(8)
    n, r, steps = 300000, 24, 32
    Z = rng.normal(size=(r, 3 * n)) * (0.85 ** np.arange(r))[:, None] * rng.uniform(0.2, 1.0, size=(1, n)).repeat(3, axis=1)
    own = (Z * Z).sum(0).reshape(n, 3).sum(1)
    E = own * (1.0 + 1e-3 * rng.uniform(0.5, 1.5, size=n))
    eng = HipEngine(0)
    scores, pred, status = _predict(eng, Z, np.ones(r), E, steps)
    assert status == 1
    sub = np.flatnonzero(scores > 0)
    print("replay subset: %d of %d vertices" % (sub.size, n))
    assert 90000 <= sub.size <= 131072
    outside = np.ones(n, bool)
    outside[sub] = False
    assert E[sub].min() >= E[outside].max()                      # the largest energies, all of them
    cols = (3 * sub[:, None] + np.arange(3)[None]).ravel()
    ms, mp = replay_model(Z[:, cols], E[sub], steps)
    mp = [int(sub[p]) for p in mp]
    agree = next((t for t in range(len(mp)) if pred[t] != mp[t]), len(mp))
    print("largest score difference %.2e, winners agree for %d of %d steps" % (np.abs(scores[sub] - ms).max(), agree, len(mp)))
    assert agree >= (3 * len(mp)) // 4 and np.abs(scores[sub] - ms).max() < 5e-4
    eng.close()


def test_replay_exchange_timeout_leaves_the_energies(monkeypatch):
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(3)
    n, r = 4000, 16
    Z = rng.normal(size=(r, 3 * n))
    E = (Z * Z).sum(0).reshape(n, 3).sum(1) * 1.01
    monkeypatch.setenv("ASB_SKETCH_TEST_STALL", "1")
    eng = HipEngine(0)
    monkeypatch.delenv("ASB_SKETCH_TEST_STALL")
    scores, pred, status = _predict(eng, Z, np.ones(r), E, 10)
    assert status == 0 and np.array_equal(scores, E)
    scores, pred, status = _predict(eng, Z, np.ones(r), E, 10)      # the stall was a one-off: the next launch runs
    assert status == 1 and scores.max() == 1.0
    eng.close()


def _param(K):
    return types.SimpleNamespace(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                                 q_support="global", vertPos_numComponents=K, store_vertPos_PCA_sing_val=False,
                                 vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, vertPos_rest_shape="first",
                                 name="t", vertPos_output_directory=".")


def _run(verts, K):
    from animsnapbases_amd import posComponents, posSnapshots
    snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False)
    comp = posComponents(_param(K), snaps)
    comp.deflate_mode = "project"
    comp.compute_components_store_singvalues()
    return comp, snaps._engine.deflate_stats()


@pytest.mark.parametrize("stall", [False, True])
def test_low_rank_data_same_basis_fewer_reads(stall, monkeypatch):
    rng = np.random.default_rng(17)
    F, N, K = 240, 40000, 96
    verts = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=40, noise=1e-4, decay=0.9, seed=17)
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)

    def check(comp):
        assert comp.selected_vertices.tolist() == d["idx"].tolist()
        comps, weigs = align_signs(comp.comps, comp.weigs, d["comps"])
        assert relerr(comps, d["comps"]) < 1e-8 and relerr(weigs, d["weigs"]) < 1e-8

    monkeypatch.setenv("ASB_SKETCH", "0")
    comp0, st0 = _run(verts, K)
    check(comp0)
    assert st0["sketch_runs"] == 0
    monkeypatch.setenv("ASB_SKETCH", "1")
    if stall:
        monkeypatch.setenv("ASB_SKETCH_TEST_STALL", "1")      # the first replay times out: that read's candidates are the plain ones
    comp1, st1 = _run(verts, K)
    check(comp1)
    assert st1["sketch_runs"] >= 1 and st1["sketch_reads"] >= 1
    reads0 = st0["panels"] + st0["refreshes"]
    reads1 = st1["panels"] + st1["refreshes"]
    print("reads of X: %d without the predictor, %d with it (%d replays)" % (reads0, reads1, st1["sketch_runs"]))
    if not stall:
        assert reads1 < reads0          # (predicted reads are taken only while they commit more per millisecond than plain ones)


def test_two_contexts_replay_at_once():
    """Two independent contexts, each on its own stream, run the same low-rank problem in two threads: their replay kernels
    (one 512-thread block per CU each: 2 x 137 blocks here, more than the chip holds at once) and panel kernels can be in flight
    together.  Whatever the scheduler does -- all blocks of a launch resident, or an exchange that times out and leaves the plain
    selection / the two-kernel loop -- both must deliver the oracle's sequence and the same basis."""
    import contextlib
    import io
    import threading
    from animsnapbases_amd import HipEngine, posComponents, posSnapshots
    rng = np.random.default_rng(23)
    F, N, K = 96, 70001, 48
    verts = orc.synth_snapshots(rng.normal(size=(N, 3)), F, rank=30, noise=1e-4, decay=0.9, seed=23)
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    outs, errs = [None, None], []

    def work(i):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False, engine=HipEngine(0))
                comp = posComponents(_param(K), snaps)
                comp.deflate_mode = "project"
                for _ in range(2):
                    comp.compute_components_store_singvalues()
            outs[i] = (comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy(), snaps._engine.deflate_stats())
        except Exception as ex:          # noqa: BLE001 -- reported by the main thread
            errs.append(repr(ex))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for idx, comps, weigs, st in outs:
        assert idx.tolist() == d["idx"].tolist()
        comps, weigs = align_signs(comps, weigs, d["comps"])
        assert relerr(comps, d["comps"]) < 1e-8 and relerr(weigs, d["weigs"]) < 1e-8
    print("replays: %d / %d, panel-kernel fallbacks: %d / %d" % (outs[0][3]["sketch_runs"], outs[1][3]["sketch_runs"],
                                                               outs[0][3]["coop_fallbacks"], outs[1][3]["coop_fallbacks"]))


@pytest.mark.parametrize("diverse", ["1", "0"])
def test_localised_bumps_same_basis_with_and_without_the_diversity_family(diverse, monkeypatch):
    """50 localised modes (Gaussian bumps, what SPLOCS is made for) at N = 40 000: the candidates of a read are half an
    energy-weighted random sample of all vertices behind a rejection (asb_project.hip: in_div) -- they only NAME candidates, so
    the selected sequence and the basis are the oracle's with the family on and off; with it the reads of X should be fewer."""
    rng = np.random.default_rng(29)
    F, N, K = 200, 40000, 80
    verts = orc.synth_snapshots(rng.uniform(size=(N, 3)), F, rank=40, noise=1e-4, decay=0.9, seed=29, kind="bumps")
    d = orc.extract_k_components(orc.prepare_snapshots(verts, "first", True)["snapTensor"], K)
    monkeypatch.setenv("ASB_DIVERSE", diverse)
    comp, st = _run(verts, K)
    assert comp.selected_vertices.tolist() == d["idx"].tolist()
    comps, weigs = align_signs(comp.comps, comp.weigs, d["comps"])
    assert relerr(comps, d["comps"]) < 1e-8 and relerr(weigs, d["weigs"]) < 1e-8
    print("ASB_DIVERSE=%s: %d reads of X (%d replays, %d refreshes)" % (diverse, st["panels"] + st["refreshes"], st["sketch_runs"],
                                                                     st["refreshes"]))
