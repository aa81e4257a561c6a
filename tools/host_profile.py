"""cProfile of the HOST side of the config-4 step (20 calls) and of config 3's second call: what the interpreter does between
the device calls.   python tools/host_profile.py"""
import os, sys, cProfile, pstats, io, types, contextlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from animsnapbases_amd import posComponents, posSnapshots
from oracle import asb_oracle as orc

def show(pr, title, n=22):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(n)
    print("=== " + title)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000], flush=True)

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(0)
X = torch.rand((2000, 100000, 3), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
with contextlib.redirect_stdout(io.StringIO()):
    snaps = posSnapshots.from_device(X.data_ptr(), 2000, 100000, rest_shape="first", standarize=True, keepalive=X)
    comp = posComponents(bench._pos_param(128, "global"), snaps)
    for _ in range(3):
        comp.extract_k_components(None)
    snaps._engine.sync()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20):
        comp.extract_k_components(None)
    snaps._engine.sync()
    pr.disable()
show(pr, "config 4: 20 steps")
del comp, snaps, X
torch.cuda.empty_cache()
rest, tris, g = bench._fixture_mesh("c3_armadillo_splocs") if os.path.exists(os.path.join(bench.ROOT, "tests", "golden", "c3_armadillo_splocs.npz")) else (None, None, None)
if rest is not None:
    F, K = int(g["F"]), int(g["param_vertPos_numComponents"])
    verts = orc.synth_snapshots(rest, F, rank=int(g["rank"]), noise=float(g["noise"]), seed=int(g["seed"]), kind=str(g["kind"]))
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False)
        snaps.compute_geodesic_distance.prepare()
        comp = posComponents(bench._pos_param(K, "local", "SPLOCS"), snaps)
        comp.compute_components_store_singvalues()
        snaps._engine.sync()
        pr = cProfile.Profile(); pr.enable()
        comp.compute_components_store_singvalues()
        snaps._engine.sync()
        pr.disable()
    show(pr, "config 3: second call", 30)
