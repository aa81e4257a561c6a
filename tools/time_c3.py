"""config 3 (bench.py's leg, no CPU baseline): ms of the compute and of the set-up, for A/B runs (ASB_ADMM_FUSED=0 ...)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
torch.cuda.set_device(0)
out = bench.other_config_pos("config3", "c3_armadillo_splocs", "local", "SPLOCS", False)
print({k: round(out[k], 2) for k in ("ms", "prepare_ms")})
