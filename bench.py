#!/usr/bin/env python3
"""bench.py -- throughput of the snapshot-reduction hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config #4, the one the north-star target is quoted on): synthetic
random snapshot tensor 100 000 vertices x 2 000 frames (float64, 4.8 GB), greedy-deflation
"PCA" with K = 128 components, global support, vertex rows sharded over the ranks.
One "step" = one complete ``extract_k_components`` over the resident tensor (F snapshots).
metric = snapshots/sec = F * steps / wall.  Inputs are resident in HBM before the timed
region (generated on the device); outputs that the reference holds in RAM (weights,
singular values, residual norms, selected vertices) are copied back inside the timed
region, the (K, N, 3) basis stays device-resident until read (DESIGN.md gives the
PCIe-inclusive figure).

Prints ONE JSON line (rank 0) with the ``roofline`` and ``cpu_baseline`` objects.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(F, N, K, budget_s=20.0):
    """Times the NumPy oracle (a port of the reference's CPU path) on a bounded sample of the
    same workload and scales it to the full job: per-component cost is linear in N (every
    operation is a streaming pass over the F x N x 3 residual)."""
    from oracle import asb_oracle as orc

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_s = max(256, min(N, N // 5))
    k_s = 2
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, size=(F, n_s, 3))
    X = orc.prepare_snapshots(X, "first", True)["snapTensor"]
    t0 = time.perf_counter()
    orc.extract_k_components(X, 1)                 # warm-up (BLAS threads, page faults)
    t_one = time.perf_counter() - t0
    k_s = int(max(2, min(K, budget_s / max(t_one, 1e-3))))
    t0 = time.perf_counter()
    orc.extract_k_components(X, k_s)
    dt = time.perf_counter() - t0
    per_comp_full = dt / k_s * (N / n_s)
    value = F / (per_comp_full * K)
    return dict(value=value, unit="snapshots/s", cores=cores, kind="port",
                sample="NumPy oracle (oracle/asb_oracle.py, OpenBLAS threads = all %d host cores) on %d of %d "
                       "vertices x %d frames, %d of %d components, %.1f s measured; scaled linearly in N and K"
                       % (cores, n_s, N, F, k_s, K, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--verts", type=int, default=100000)
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--comps", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    # native libraries (RCCL prints a version banner) write to fd 1: keep the real stdout for the ONE JSON line
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ASB_FORCE_COLLECTIVES=1 (with --gpus 1): single-rank RCCL group, multi-rank protocol -- measures what the
    # per-panel collectives and their host synchronisation cost on top of the kernels
    forced = os.environ.get("ASB_FORCE_COLLECTIVES", "0") == "1"
    if world > 1 or forced:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from animsnapbases_amd import Comm, partition, posComponents, posSnapshots

    F, N, K = args.frames, args.verts, args.comps
    comm = Comm()
    v0, n_loc = partition(N, world)[rank]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    Xd = torch.rand((F, n_loc, 3), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    torch.cuda.synchronize()

    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
        snaps = posSnapshots.from_device(Xd.data_ptr(), F, n_loc, rest_shape="first", standarize=True, comm=comm,
                                         keepalive=Xd)
    del Xd
    torch.cuda.empty_cache()
    param = types.SimpleNamespace(vertPos_bases_type="PCA", vertPos_numComponents=K, q_support="global",
                                  store_vertPos_PCA_sing_val=False, vertPos_smooth_min_dist=0.1,
                                  vertPos_smooth_max_dist=0.25, q_standarize=True, q_massWeight=False,
                                  q_orthogonal=False, vertPos_output_directory=".", name="bench")
    comp = posComponents(param, snaps)
    eng = snaps._engine

    def step():
        with contextlib.redirect_stdout(quiet):
            comp.extract_k_components(None)

    def fence():
        torch.cuda.synchronize()
        eng.sync()
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.prof_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    launches, kern_ms = eng.prof_get()
    eng.prof_reset(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity of the result that was just timed (not a parity test: see tests/)
    assert len(set(comp.selected_vertices.tolist())) == K
    assert np.isfinite(comp.measures_at_largeDeforVerts).all()

    if rank == 0:
        value = F * args.steps / dt
        # dominant kernel: one streaming pass over this rank's shard -- k_project_mfma (projection mode:
        # one launch per PANEL of up to 16 components) or k_stream (residual mode: one per component).
        # algorithmic bytes per launch = 24 * n_loc * F (c = 1: one read of the shard)
        stats = eng.deflate_stats()
        # HBM traffic from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, corrected
        # as the microarch guide prescribes; tools/summarise_pmc.py -> profiles/*_pmc_traffic.json).  Per pass over
        # the shard: k_project_lds takes ceil(F/1008) sweep launches per pass.
        traffic = None
        pk = int(os.environ.get("ASB_PROJECT_KERNEL", "3"))
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01h_pmc_traffic.json")))["kernels"]
            kname = {2: "k_project_lds", 3: "k_project_l2s<4, 2, 2, 1>"}.get(pk)
            if getattr(eng, "mode", 0) == 1 and kname in pm and (N, F, world) == (100000, 2000, 1) and \
                    os.environ.get("ASB_SUPER_PANELS", "0") != "1" and os.environ.get("ASB_L2_VARIANT", "4") == "4":
                traffic = pm[kname]["hbm_bytes"]
        except Exception:
            traffic = None
        mode = "project" if getattr(eng, "mode", 0) == 1 else "residual"
        nsweep = -(-((F + 15) // 16) // 63) if (getattr(eng, "mode", 0) == 1 and pk == 2) else 1   # k_project_lds: sweeps per pass
        alg_bytes = 24.0 * n_loc * F / nsweep
        avg_ms = kern_ms / max(launches, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if launches else None
        out = {
            "metric": "snapshots/sec (SVD+SPLOCS) for n_verts x n_frames; basis Frobenius err vs ref",
            "value": value, "unit": "snapshots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config4: synthetic U[-1,1) %d verts x %d frames, greedy-deflation PCA K=%d, "
                                   "global support, standardised, vertex rows sharded over %d GPU(s)" % (N, F, K, world),
                       "n_verts": N, "n_frames": F, "K": K, "parallelism": "vertex-shard x%d" % world + (" (multi-rank protocol forced)" if forced else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": (("k_project_wide<NT,G,1..3> (super-panels: up to 48 columns per read of X)"
                                     if os.environ.get("ASB_SUPER_PANELS", "0") == "1" and mode == "project" else
                                     {2: "k_project_lds", 3: "k_project_l2s<4,2,2,1>"}.get(pk, "k_project_mfma")) +
                                    " (f64-MFMA panel projection, %d launch(es) = one read of X per panel)" % nsweep
                                    if mode == "project" else "k_stream<T,E2,UPDATE> (deflation pass, read+write of R)"),
                         "algorithm": mode, "panels_per_step": stats["panels"], "refreshes": stats["refreshes"],
                         "launches": launches,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         # SURVEY.md 8(d) prices a whole step at 24 N F (1 + K) bytes (one read of X per component);
                         # the panel algorithm reads X once per PANEL, so the step beats that figure's own roofline
                         "survey_step_bytes": 24.0 * N * F * (1 + K),
                         "survey_step_equivalent_GBps": 24.0 * N * F * (1 + K) / (dt / args.steps) / 1e9},
        }
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would wait for it)
            out["cpu_baseline"] = cpu_baseline(F, N, K, args.cpu_budget)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
