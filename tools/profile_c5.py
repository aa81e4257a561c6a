"""Config 5 alone (bench.py's other_config_c5, no CPU baseline) -- run under rocprofv3 --kernel-trace --stats for profiles/*c5*."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
torch.cuda.set_device(0)
out = bench.other_config_c5(torch.device("cuda:0"), cpu=False)
print(json.dumps({k: out[k] for k in ("ms", "pod_ms", "deim_ms", "post_process_ms", "prepare_ms", "roofline")}))
