"""One FULL (un-sampled) run of config 4 on the host cores: the NumPy oracle's greedy deflation on 2000 frames x 100 000 vertices,
K = 128 -- the anchor of bench.py's extrapolated `cpu_baseline` (which times a tenth of the vertices and a few components and
scales linearly).  The oracle's loop carries no state but the residual, so it is called in chunks of 32 components on the residual
the previous chunk returned (three extra 4.8 GB copies, < 1 % of the run): a run cut short still leaves per-chunk times behind.
Then the HIP path on the SAME tensor, compared with the oracle's output at full size.

    python tools/cpu_full_config4.py OUT.json [N] [K]
"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import asb_oracle as orc
import bench

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cpu_full_config4.json"
F, N, K = 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 100000, int(sys.argv[3]) if len(sys.argv) > 3 else 128
CH = 32
cores, model = bench._cpu_info()
t_start = time.perf_counter()
stop = threading.Event()


def heartbeat():
    while not stop.wait(60.0):
        print("  ... %.0f s" % (time.perf_counter() - t_start), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
rng = np.random.default_rng(7)
raw = rng.uniform(-1, 1, size=(F, N, 3))
print("tensor drawn: %.1f s" % (time.perf_counter() - t_start), flush=True)
rec = dict(workload="config 4: %d frames x %d vertices x 3, K = %d, f64, uniform(-1, 1) seed 7, rest shape = first frame, standardised"
                    % (F, N, K), cores=cores, cpu_model=model, chunk=CH, chunks_s=[])
t0 = time.perf_counter()
X = orc.prepare_snapshots(raw, "first", True)["snapTensor"]
rec["prepare_s"] = time.perf_counter() - t0
print("prepared: %.1f s" % rec["prepare_s"], flush=True)
R, comps, weigs, idx = X, [], [], []
for c0 in range(0, K, CH):
    t0 = time.perf_counter()
    r = orc.extract_k_components(R, min(CH, K - c0))
    dt = time.perf_counter() - t0
    R = r["R"]
    comps.append(r["comps"]); weigs.append(r["weigs"]); idx.append(r["idx"])
    rec["chunks_s"].append(round(dt, 2))
    rec["total_s"] = round(sum(rec["chunks_s"]), 2)
    rec["components_done"] = c0 + r["idx"].shape[0]
    rec["snapshots_per_s"] = F / (rec["total_s"] * K / rec["components_done"])
    print("components %d: chunk %.1f s, total %.1f s" % (rec["components_done"], dt, rec["total_s"]), flush=True)
    json.dump(rec, open(out, "w"), indent=1)
comps, weigs, idx = np.concatenate(comps), np.concatenate(weigs, axis=1), np.concatenate(idx)
resid = float(np.linalg.norm(R))
del R
rec["residual_norm"] = resid

# the HIP path on the same tensor
import types
from animsnapbases_amd import posComponents, posSnapshots
tris = np.zeros((1, 3), dtype=np.int64)
t0 = time.perf_counter()
snaps = posSnapshots.from_arrays(raw, tris, "first", standarize=True, massWeight=False)
comp = posComponents(bench._pos_param(K, "global"), snaps)
bench._quiet(comp.compute_components_store_singvalues)
snaps._engine.sync()
t1 = time.perf_counter()
bench._quiet(comp.compute_components_store_singvalues)
snaps._engine.sync()
rec["hip_second_call_ms"] = (time.perf_counter() - t1) * 1e3
C = np.asarray(comp.comps)
W = np.asarray(comp.weigs)
gi = np.asarray(comp.selected_vertices)
sgn = np.sign(np.einsum("fk,fk->k", W, weigs))          # LAPACK's sign of a singular vector is arbitrary: align per component
sgn[sgn == 0] = 1.0
rec["hip_vs_oracle"] = dict(
    comps_rel=float(np.linalg.norm(C * sgn[:, None, None] - comps) / np.linalg.norm(comps)),
    comps_max_abs=float(np.abs(C * sgn[:, None, None] - comps).max()),
    weigs_rel=float(np.linalg.norm(W * sgn[None] - weigs) / np.linalg.norm(weigs)),
    picked_equal=bool(np.array_equal(gi.ravel(), idx)), sign_aligned=True)
stop.set()
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k != "chunks_s"}), flush=True)
