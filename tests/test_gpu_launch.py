"""GPU (-m gpu): bench.py --gpus 2 started the way the driver starts it -- a fresh child process running
``python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`` -- on the ONE-GPU box: both ranks on device 0
(ASB_BENCH_ONE_DEVICE=1), a gloo group whose collectives are staged through the host (ASB_BENCH_BACKEND=gloo,
animsnapbases_amd/distributed.py).  What it rehearses before a multi-GPU box ever sees it: RANK / LOCAL_RANK / WORLD_SIZE from
the launcher, per-rank seeds and vertex partition, the multi-rank panel protocol between two PROCESSES (each with its own
context and co-resident kernels on the shared GPU: an exchange that times out must end in the collective fall-back, not a hang),
the barriers around the timed region, the teardown, and the ONE JSON line on rank 0's stdout.  The numbers mean nothing.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2])
def test_bench_under_the_launcher_two_processes_one_gpu(world):
    env = dict(os.environ, ASB_BENCH_ONE_DEVICE="1", ASB_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--verts", "60000", "--frames", "256", "--comps", "64", "--no-cpu-baseline", "--no-other-configs"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout.decode()[-2000:], err[-2000:])            # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "snapshots/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["n_verts"] == 60000 and d["config"]["K"] == 64
    assert "LAUNCH REHEARSAL" in d["config"]["parallelism"]
    assert d["scaling"] == "strong" and d["dtype"] == "f64"
    print("2-process launch on one GPU: %.1f ms per step, %d panel-kernel fallbacks" %
          (d["ms_per_step"], d["roofline"]["panel_kernel_fallbacks"]))
