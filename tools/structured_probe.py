"""Reads of X per K = 128 components on several kinds of STRUCTURED data of config 4's shape, with the sketch predictor on and off
(ASB_SKETCH): iid low-rank modes (bench.py's leg), a slowly decaying spectrum, localised bumps (what SPLOCS is made for).

    python tools/structured_probe.py [kind ...]        # kinds: lowrank slow bumps
"""
import contextlib, io, os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

N, F, K = 100000, 2000, 128
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def make(kind, seed=5):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    f64 = dict(dtype=torch.float64, device=dev)
    rest = torch.rand((N, 3), generator=gen, **f64)
    if kind == "bumps":
        r = 50
        modes = torch.empty((r, N, 3), **f64)
        for j in range(r):
            c = rest[int(torch.randint(N, (1,), generator=gen, device=dev))]
            rad = 0.1 + 0.25 * float(torch.rand((1,), generator=gen, **f64))
            d = torch.randn((3,), generator=gen, **f64)
            modes[j] = torch.exp(-((rest - c) ** 2).sum(1) / rad ** 2)[:, None] * (d / d.norm())[None] * 0.1
        coef = torch.randn((F, r), generator=gen, **f64) * (0.9 ** torch.arange(r, **f64))[None]
        X = rest.reshape(1, -1) + coef @ modes.reshape(r, -1)
    else:
        r, decay = (50, 0.9) if kind == "lowrank" else (200, 0.97)
        coef = torch.randn((F, r), generator=gen, **f64) * (decay ** torch.arange(r, **f64))[None]
        modes = 0.02 * torch.randn((r, N * 3), generator=gen, **f64)
        X = rest.reshape(1, -1) + coef @ modes
    X += 1e-4 * torch.randn((F, N * 3), generator=gen, **f64)
    return X.reshape(F, N, 3).contiguous()


for kind in (sys.argv[1:] or ["lowrank", "slow", "bumps"]):
    Xd = make(kind)
    torch.cuda.synchronize()
    from animsnapbases_amd import posComponents, posSnapshots
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_device(Xd.data_ptr(), F, N, rest_shape="first", standarize=True, keepalive=Xd)
        comp = posComponents(bench._pos_param(K, "global"), snaps)
        comp.extract_k_components(None)
        snaps._engine.sync()
        t0 = time.perf_counter()
        comp.extract_k_components(None)
        snaps._engine.sync()
        ms = (time.perf_counter() - t0) * 1e3
    st = snaps._engine.deflate_stats()
    print("%-8s ASB_SKETCH=%s: %.1f ms, %d reads of X (%d replays)%s" % (kind, os.environ.get("ASB_SKETCH", "1"), ms,
          st["panels"] + st["refreshes"], st["sketch_runs"],
          "  !! %d panel-kernel fallbacks, %d refreshes" % (st["coop_fallbacks"], st["refreshes"]) if st["coop_fallbacks"] or st["refreshes"] else ""),
          flush=True)
    del comp, snaps, Xd
    torch.cuda.empty_cache()
