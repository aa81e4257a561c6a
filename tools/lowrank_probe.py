"""Reads of X per component on structured data (bench.py's c4_lowrank leg), with the per-panel log (ASB_DEBUG_PANELS=1)."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
t0 = time.time()
out = bench.other_config_c4_lowrank(dev, seed=int(os.environ.get("LR_SEED", "77")))
print({k: out[k] for k in ("ms", "reads_of_X", "panels", "refreshes")}, "wall %.1f s" % (time.time() - t0))
