"""Where config 5's DEIM milliseconds go on the host side: cProfile of constraintsComponents.deim (second cycle of the bench leg)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from animsnapbases_amd import constraints

torch.cuda.set_device(0)
orig = constraints.constraintsComponents.deim
calls = {"n": 0}
def deim(self):
    calls["n"] += 1
    if calls["n"] % 2:
        return orig(self)
    pr = cProfile.Profile()
    pr.enable()
    r = orig(self)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    sys.stderr.write(s.getvalue())
    return r
constraints.constraintsComponents.deim = deim
out = bench.other_config_c5(torch.device("cuda", 0), cpu=False)
print("deim_ms", round(out["deim_ms"], 2))
