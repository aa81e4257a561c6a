"""Reads of X per K = 128 components on several kinds of STRUCTURED data of config 4's shape, with the sketch predictor on and off
(ASB_SKETCH): iid low-rank modes (bench.py's leg), a slowly decaying spectrum, localised bumps (what SPLOCS is made for).

    python tools/structured_probe.py [kind ...]        # kinds: lowrank slow bumps rankdef   (PROBE_MODE=residual|project forces the device algorithm)
"""
import contextlib, io, os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

N, F, K = 100000, 2000, 128
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def make(kind, seed=5):
    return bench.structured_tensor(dev, {"slow": "slow_spectrum", "rankdef": "rank_deficient"}.get(kind, kind), N, F, seed).contiguous()


for kind in (sys.argv[1:] or ["lowrank", "slow", "bumps"]):
    Xd = make(kind)
    torch.cuda.synchronize()
    from animsnapbases_amd import posComponents, posSnapshots
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_device(Xd.data_ptr(), F, N, rest_shape="first", standarize=True, keepalive=Xd)
        comp = posComponents(bench._pos_param(K, "global"), snaps)
        if os.environ.get("PROBE_MODE"):
            comp.deflate_mode = os.environ["PROBE_MODE"]
        comp.extract_k_components(None)
        snaps._engine.sync()
        t0 = time.perf_counter()
        comp.extract_k_components(None)
        snaps._engine.sync()
        ms = (time.perf_counter() - t0) * 1e3
    st = snaps._engine.deflate_stats()
    print("%-8s ASB_SKETCH=%s: %.1f ms, %d reads of X (%d replays)%s" % (kind, os.environ.get("ASB_SKETCH", "1"), ms,
          st["panels"] + st["refreshes"], st["sketch_runs"],
          ("  !! %d panel-kernel fallbacks, %d refreshes" % (st["coop_fallbacks"], st["refreshes"]) if st["coop_fallbacks"] or st["refreshes"] else "")
          + ("  -> residual loop from component %d" % st["residual_switch_at"] if st.get("residual_switch_at", -1) >= 0 else "")),
          flush=True)
    del comp, snaps, Xd
    torch.cuda.empty_cache()
