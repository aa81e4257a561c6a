"""Long seeded sweeps of the other paths (not part of the test suite):
   python tools/fuzz_sweep_more.py multirank|local|constraints|linalg [first_seed] [count] [--big]"""
import contextlib, io, sys, time, types
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from animsnapbases_amd import HipEngine, posComponents, posSnapshots, constraintsComponents, nonlinearSnapshots
from oracle import asb_oracle as orc
from conftest import align_signs, relerr

what = sys.argv[1]
first, count = int(sys.argv[2]) if len(sys.argv) > 2 else 5000, int(sys.argv[3]) if len(sys.argv) > 3 else 40
bad, t0 = 0, time.time()


def pparam(**over):
    base = dict(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False, q_support="global",
                vertPos_numComponents=4, store_vertPos_PCA_sing_val=False, vertPos_smooth_min_dist=0.1,
                vertPos_smooth_max_dist=0.25, vertPos_rest_shape="first", name="t", vertPos_output_directory=".")
    base.update(over)
    return types.SimpleNamespace(**base)


def report(seed, msg):
    global bad
    bad += 1
    print("seed", seed, msg, flush=True)


for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    try:
        if what == "multirank":
            from thread_comm import run_ranks
            world = int(rng.integers(2, 5))
            N, F = int(rng.integers(world, 4000)), int(rng.integers(2, 300))
            K = int(max(1, min(rng.integers(1, 40), (min(F - 1, 3 * N) + 1) // 2)))
            mode = ["project", "residual"][seed % 2]
            if "--big" in sys.argv:               # shards above the candidate capacity: guessed first panel, several sub-panels per read
                N, F = int(rng.integers(4000, 30000)), int(rng.integers(40, 300))
                K = int(max(1, min(rng.integers(20, 90), (F - 1) // 2)))
                mode = "project"
            rest = str(rng.choice(["first", "average"]))
            verts = rng.uniform(-1, 1, size=(F, N, 3)) * rng.uniform(0.2, 1, size=(1, N, 1))
            param = pparam(vertPos_numComponents=K, vertPos_rest_shape=rest)

            def rank_fn(rank, comm):
                snaps = posSnapshots.from_arrays(verts, None, rest, standarize=True, massWeight=False,
                                                 engine=HipEngine(0, stream=0), comm=comm)
                comp = posComponents(param, snaps)
                comp.deflate_mode = mode
                comp.compute_components_store_singvalues()
                return comp.selected_vertices.copy(), comp.comps.copy(), comp.weigs.copy()
            with contextlib.redirect_stdout(io.StringIO()):        # once, in the main thread (the swap is process-wide)
                outs = run_ranks(world, rank_fn)
            pre = orc.prepare_snapshots(verts, rest, True)
            ref = orc.extract_k_components(pre["snapTensor"], K)
            sig = ref["measures"][:, 1]
            good = int(np.argmax(sig < 1e-9 * sig[0])) if np.any(sig < 1e-9 * sig[0]) else K
            for idx, comps, weigs in outs:
                if idx[:good].tolist() != ref["idx"][:good].tolist():
                    report(seed, ("idx", world, verts.shape, K, mode)); break
                c, w = align_signs(comps[:good], weigs[:, :good], ref["comps"][:good])
                if not max(relerr(c, ref["comps"][:good]), relerr(w, ref["weigs"][:, :good])) < 1e-7:
                    report(seed, ("val", world, verts.shape, K, mode)); break
        elif what == "local":
            rings, segs = int(rng.integers(3, 30)), int(rng.integers(4, 40))
            F = int(rng.integers(6, 200))
            rest_v, tris = orc.synth_mesh(rings, segs, seed=seed)
            N = rest_v.shape[0]
            K = int(max(1, min(rng.integers(2, 24), F // 2, N // 2)))
            verts = orc.synth_snapshots(rest_v, F, rank=K + 4, seed=seed, kind="bumps", decay=0.93)
            dmax = float(rng.uniform(0.2, 0.7))
            splocs = seed % 3 == 0
            param = pparam(vertPos_numComponents=K, q_support="local", vertPos_smooth_max_dist=dmax,
                           vertPos_bases_type="SPLOCS" if splocs else "PCA", splocs_max_itrs=3, splocs_admm_num_itrs=4,
                           splocs_lambda=2.0, splocs_rho=10.0)
            with contextlib.redirect_stdout(io.StringIO()):
                snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False)
                comp = posComponents(param, snaps)
                comp.compute_components_store_singvalues()
            pre = orc.prepare_snapshots(verts, "first", True)
            geo = orc.Geodesics(verts[0], tris)
            ref = orc.extract_k_components(pre["snapTensor"], K, "local", geo, 0.1, dmax)
            if comp.selected_vertices.tolist() != ref["idx"].tolist():
                report(seed, ("idx", rings, segs, F, K))
            elif not max(relerr(comp.comps, ref["comps"]), relerr(comp.weigs, ref["weigs"])) < 1e-6:
                report(seed, ("val", rings, segs, F, K, relerr(comp.comps, ref["comps"])))
            elif splocs:
                s = orc.splocs_glob_optimization(pre["snapTensor"], ref["comps"], ref["weigs"], ref["R"], geo, 0.1, dmax, 3, 4, 2.0, 10.0)
                if comp.splocs_centres.tolist() != s["idx"].tolist() or not np.allclose(comp.splocs_trace, s["trace"], rtol=1e-7):
                    report(seed, ("splocs", rings, segs, F, K))
        elif what == "constraints":
            p = int(rng.choice([1, 1, 2, 3]))
            e_, F = int(rng.integers(4, 900)), int(rng.integers(3, 200))
            ep = e_ * p
            r = int(rng.integers(2, 30))
            frames = 0.2 + np.tensordot(rng.normal(size=(F, r)) * (0.9 ** np.arange(r))[None], rng.normal(size=(r, ep, 3)), (1, 0)) \
                + 1e-5 * rng.normal(size=(F, ep, 3))
            kindb = ["pod_vectorized", "pca_blocks"][seed % 2]
            K = int(max(1, min(rng.integers(1, 30), (F - 1) // (2 * p) if kindb == "pca_blocks" else min(F, 3 * ep))))
            K = max(K, 1)
            param = types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=p,
                                          constProj_massWeight=False, constProj_standarize=True, constProj_orthogonal=bool(seed % 3 == 0),
                                          constProj_basis_type=kindb, deim_desired_num_components=K,
                                          constProj_store_sing_val=False, constProj_output_directory=".", name="c5", constProj_name="v")
            pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
            if kindb == "pod_vectorized":
                if 3 * ep < F:
                    continue
                s_all = np.linalg.svd(pre["snapTensor"].reshape(F, -1), compute_uv=False)
                if not s_all[K - 1] > 1e-6 * s_all[0]:
                    continue                               # the product refuses numerically rank-deficient requests
            with contextlib.redirect_stdout(io.StringIO()):
                ns = nonlinearSnapshots(param, frames=frames); ns.config(); ns.snapshots_prepare()
                cc = constraintsComponents(param, ns); cc.config(); cc.compute_components_store_singvalues()
            if kindb == "pca_blocks":
                rr = orc.pca_blocks(pre["snapTensor"], K, p)
                sgn = np.sign(np.sum(cc.weigs * rr["weigs"], axis=0))
                if cc.largeDeforBlocks.tolist() != rr["blocks"].tolist():
                    report(seed, ("blocks idx", frames.shape, K, p))
                elif not relerr(cc.comps * sgn[:, None, None], rr["comps"]) < 1e-7:
                    report(seed, ("blocks val", frames.shape, K, p, relerr(cc.comps * sgn[:, None, None], rr["comps"])))
            else:
                pod = orc.pod_vectorized(pre["snapTensor"], K)
                keep = pod["S"][:K] > 3e-7 * pod["S"][0]          # refined route: vector error ~ eps s0 / sk
                got, want = cc.comps.reshape(K, -1)[keep], pod["comps"].reshape(K, -1)[keep]
                sg = np.sign(np.sum(got * want, axis=1))
                if not relerr(cc.singular_values[:K][keep], pod["S"][:K][keep]) < 1e-8 or not relerr(got * sg[:, None], want) < 1e-5:
                    report(seed, ("pod", frames.shape, K, relerr(got * sg[:, None], want)))
                with contextlib.redirect_stdout(io.StringIO()):
                    cc.post_process_components(); cc.deim()
                if len(set(cc.geom_Pt.tolist())) != K:
                    report(seed, ("deim points", frames.shape, K))
        elif what == "linalg":
            import torch
            from scipy.linalg import eigh_tridiagonal
            from animsnapbases_amd._lib import ptr
            n = int(rng.integers(1, 900))
            B = rng.normal(size=(n, int(rng.integers(1, n + 1))))
            A = B @ B.T + 10.0 ** rng.integers(-3, 1) * n * np.eye(n)
            e = HipEngine(0, stream=0)
            out = np.empty((n, n))
            e._ck(e.lib.asb_test_spd_inverse(e.h, ptr(np.ascontiguousarray(A)), n, ptr(out)))
            ref_inv = np.linalg.inv(A)
            if not np.abs(out - ref_inv).max() < 1e-13 * np.linalg.cond(A) * np.abs(ref_inv).max():
                report(seed, ("inverse", n, np.abs(out - ref_inv).max() / np.abs(ref_inv).max(), np.linalg.cond(A)))
            Ad = torch.from_numpy(A.copy()).cuda()
            d, off = e.sym_tridiag(n, Ad.data_ptr())
            lam_ref = np.linalg.eigvalsh(A)
            lam = eigh_tridiagonal(d, off, eigvals_only=True) if n > 1 else d
            if not np.allclose(lam, lam_ref, rtol=0, atol=1e-11 * abs(lam_ref).max()):
                report(seed, ("tridiag", n, np.abs(lam - lam_ref).max() / abs(lam_ref).max()))
            e.close()
    except Exception as ex:
        report(seed, ("RAISED", repr(ex)[:300]))
    if (seed - first) % 10 == 9:
        print("... %d cases, %d bad, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
print("done:", what, count, "cases,", bad, "bad")
