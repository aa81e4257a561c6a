"""Long seeded sweep (not part of the test suite): random shapes / data kinds, both device algorithms against the
oracle.  python tools/fuzz_sweep.py [first_seed] [count] [--big] [--bigk] [--prep]
--big: shards of 20 000 .. 70 000 vertices (several sub-panels per read of X); --prep: through posSnapshots / posComponents
(rest shape, standardisation sweep: the guessed first panel) instead of a bare upload."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from animsnapbases_amd import HipEngine
from oracle import asb_oracle as orc
from conftest import align_signs, relerr

nums = [a for a in sys.argv[1:] if not a.startswith("-")]
first, count = int(nums[0]) if len(nums) > 0 else 1000, int(nums[1]) if len(nums) > 1 else 100
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    big = "--big" in sys.argv                 # shards large enough for super-panels (ASB_SUPER_PANELS=1)
    N = int(rng.integers(20000, 70000)) if big else int(rng.integers(1, 6000))
    F = int(rng.integers(8, 150)) if big else int(rng.integers(1, 700))
    if "--bigk" in sys.argv:                  # reads with four sub-panels (K >= 64): odd frame counts, ragged shards
        F = int(rng.integers(130, 330))
    kind = rng.choice(["uniform", "lowrank", "dupes", "zeros", "scaled", "smooth"])
    if kind == "lowrank":
        r = int(rng.integers(1, 20))
        X = (rng.normal(size=(F, r)) * (rng.uniform(0.5, 0.95) ** np.arange(r))) @ rng.normal(size=(r, N * 3))
        X = X.reshape(F, N, 3) + 10.0 ** rng.integers(-9, -3) * rng.normal(size=(F, N, 3))
    elif kind == "smooth":
        t = np.linspace(0, 1, F)[:, None, None]
        X = np.sin(2 * np.pi * (t * rng.uniform(0.5, 4, size=(1, N, 3)) + rng.uniform(size=(1, N, 3)))) * rng.uniform(0.1, 1, size=(1, N, 1))
    else:
        X = rng.uniform(-1, 1, size=(F, N, 3))
    if kind == "dupes" and N > 4:
        for _ in range(int(rng.integers(1, 6))):
            a, b = rng.integers(0, N, size=2)
            X[:, a] = X[:, b]
    if kind == "zeros" and N > 3:
        X[:, rng.integers(0, N, size=N // 3)] = 0.0
    if kind == "scaled":
        X *= 10.0 ** rng.integers(-140, 140)
    K = int(max(1, min(rng.integers(1, 70), (min(F, 3 * N) + 1) // 2)))
    if "--bigk" in sys.argv:
        K = int(max(1, min(rng.integers(60, 150), (min(F, 3 * N) + 1) // 2)))
    if "--prep" in sys.argv:
        import contextlib, io, types
        from animsnapbases_amd import posComponents, posSnapshots
        rest = ["first", "average"][seed % 2]
        if kind == "scaled":
            X = X / np.abs(X).max()               # (the reference's np.std overflows on 1e140)
        X = X + rng.uniform(-1, 1, size=(1, N, 3)) * float(np.abs(X).max() + 1e-300) * (seed % 3)      # per-vertex offsets of varying weight
        pre = orc.prepare_snapshots(X, rest, True)
        if not np.all(np.isfinite(pre["snapTensor"])) or F < 3:
            continue
        K = int(max(1, min(K, (min(F - 1, 3 * N) + 1) // 2)))
        ref = orc.extract_k_components(pre["snapTensor"], K)
        if not np.all(np.isfinite(ref["comps"])):
            continue
        sig = ref["measures"][:, 1]
        good = int(np.argmax(sig < 1e-9 * sig[0])) if np.any(sig < 1e-9 * sig[0]) else K
        param = types.SimpleNamespace(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                                      q_support="global", vertPos_numComponents=K, store_vertPos_PCA_sing_val=False,
                                      vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, vertPos_rest_shape=rest, name="t",
                                      vertPos_output_directory=".")
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                snaps = posSnapshots.from_arrays(X, None, rest, standarize=True, massWeight=False)
                comp = posComponents(param, snaps)
                comp.deflate_mode = "project"
                comp.extract_k_components(None)
            st = snaps._engine.deflate_stats()
            ok_idx = comp.selected_vertices[:good].tolist() == ref["idx"][:good].tolist()
            err = np.inf
            if ok_idx and good:
                comps, weigs = align_signs(comp.comps[:good], comp.weigs[:, :good], ref["comps"][:good])
                err = max(relerr(comps, ref["comps"][:good]), relerr(weigs, ref["weigs"][:, :good]))
            if not ok_idx or not err < 1e-7:
                bad += 1
                print("seed", seed, kind, rest, X.shape, K, "good", good, "idx_ok", ok_idx, "err %.2e" % err, st, flush=True)
            elif "-v" in sys.argv:
                print("seed", seed, kind, rest, X.shape, K, "ok %.1e" % err, st, flush=True)
        except Exception as ex:
            bad += 1
            print("seed", seed, kind, rest, X.shape, K, "RAISED", repr(ex)[:200], flush=True)
        if (seed - first) % 20 == 19:
            print("... %d cases, %d bad, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
        continue
    ref = orc.extract_k_components(X, K)
    if not np.all(np.isfinite(ref["comps"])):
        continue
    sig = ref["measures"][:, 1]
    good = int(np.argmax(sig < 1e-9 * sig[0])) if np.any(sig < 1e-9 * sig[0]) else K
    for mode in ((1,) if big else (0, 1)):
        if "-v" in sys.argv:
            print("start seed", seed, kind, X.shape, "K", K, "mode", mode, flush=True)
        e = HipEngine(0)
        e.upload(X, 0, N)
        e.deflate_begin(K, False, mode)
        try:
            e.run_global(0, K)
            r = e.results()
        except Exception as ex:
            print("seed", seed, kind, X.shape, K, "mode", mode, "RAISED", repr(ex)[:200], flush=True)
            bad += 1
            e.close()
            continue
        e.close()
        ok_idx = r["idx"][:good].tolist() == ref["idx"][:good].tolist()
        err = np.inf
        if ok_idx and good:
            comps, weigs = align_signs(r["comps"][:good], r["weigs"][:, :good], ref["comps"][:good])
            err = max(relerr(comps, ref["comps"][:good]), relerr(weigs, ref["weigs"][:, :good]))
        if not ok_idx or not err < 1e-7:
            bad += 1
            print("seed", seed, kind, X.shape, K, "good", good, "mode", mode, "idx_ok", ok_idx, "err %.2e" % err, flush=True)
    if (seed - first) % 20 == 19:
        print("... %d cases, %d bad, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
print("done:", count, "cases,", bad, "bad")
