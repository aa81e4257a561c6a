"""config 5 (bench.py's leg, no CPU baseline), three times: POD / post-process / DEIM ms."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = bench.other_config_c5(dev, cpu=False)
    print({k: round(out[k], 2) for k in ("ms", "pod_ms", "post_process_ms", "deim_ms", "prepare_ms")}, flush=True)
