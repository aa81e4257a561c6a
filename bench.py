#!/usr/bin/env python3
"""bench.py -- throughput of the snapshot-reduction hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5        (the defaults)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json config #4, the one the north-star target is quoted on): synthetic random snapshot
tensor 100 000 vertices x 2 000 frames (float64, 4.8 GB), greedy-deflation "PCA" with K = 128 components, global
support, vertex rows sharded over the ranks.  One "step" = one complete ``extract_k_components`` over the resident
tensor (F snapshots).  metric = snapshots/sec = F * steps / wall.  Inputs are resident in HBM before the timed region
(generated on the device); what the reference holds in RAM afterwards (weights, singular values, residual norms,
selected vertices) is copied back inside the timed region; the (K, N, 3) basis stays device-resident until it is read
-- ``end_to_end`` adds the preparation (layout change + standardisation, ``prepare_ms``) and that download.

Prints ONE JSON line (rank 0).  Besides the contract's keys:
  roofline      dominant kernel (HIP events around every launch of it on the engine's stream) AND the step-level
                figure: (reads of X per step) x 24 N F bytes / ms_per_step against 8 TB/s;
  cpu_baseline  the NumPy oracle (a port of the reference's CPU path) on a bounded sample, median of 3;
  end_to_end    prepare_ms + ms_per_step + basis_download_ms;
  other_configs BASELINE.json configs 2, 3 and 5 at their full size (one GPU), each with its own time, roofline and CPU
                baseline (N = 1 only; ``--no-other-configs`` skips them).
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
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (vendor spec; the guide lists no f64 row; tools/probe_mfma_f64: 70 measured)
PMC_PROFILE = os.path.join("profiles", "r04e_pmc_traffic.json")      # offline rocprofv3 --pmc passes of THIS build


def _cpu_info():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, model


def _median3(fn):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def cpu_baseline_c4(F, N, K, budget_s=18.0):
    """NumPy oracle on a bounded sample of config 4, scaled: the per-component cost is linear in N (every operation is a
    streaming pass over the F x N x 3 residual) and the loop is linear in K.  Median of 3 runs of the sample."""
    from oracle import asb_oracle as orc
    cores, model = _cpu_info()
    n_s = max(256, min(N, N // 4))          # (a tenth of the vertices under-rated the host by 1.46x: profiles/r04_cpu_full_config4.json)
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, size=(F, n_s, 3))
    X = orc.prepare_snapshots(X, "first", True)["snapTensor"]
    t0 = time.perf_counter()
    orc.extract_k_components(X, 1)                 # warm-up (BLAS threads, page faults)
    t_one = time.perf_counter() - t0
    k_s = int(max(2, min(K, budget_s / 3.0 / max(t_one, 1e-3))))
    med, ts = _median3(lambda: orc.extract_k_components(X, k_s))
    per_comp_full = med / k_s * (N / n_s)
    anchor = None
    try:
        a = json.load(open(os.path.join(ROOT, "profiles", "r04_cpu_full_config4.json")))
        anchor = {"snapshots_per_s": a["snapshots_per_s"], "total_s": a["total_s"], "cores": a["cores"], "cpu_model": a["cpu_model"],
                  "file": "profiles/r04_cpu_full_config4.json", "note": "ONE full un-sampled run of the same oracle on config 4 (recorded, not this run)"}
    except (OSError, ValueError, KeyError):
        pass
    return dict(value=F / (per_comp_full * K), unit="snapshots/s", cores=cores, kind="port", cpu_model=model, full_run_recorded=anchor,
                runs_s=[round(t, 3) for t in ts],
                sample="NumPy oracle (oracle/asb_oracle.py, OpenBLAS threads = all %d host cores) on %d of %d vertices x %d "
                       "frames, %d of %d components; median of 3 runs (%.2f s); scaled linearly in N and K"
                       % (cores, n_s, N, F, k_s, K, med))


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs 2, 3, 5 (one GPU, full size)
# ---------------------------------------------------------------------------------------------------------------------
def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _timed(fn, sync):
    sync()
    t0 = time.perf_counter()
    r = _quiet(fn)
    sync()
    return (time.perf_counter() - t0) * 1e3, r


def _pos_param(K, support, kind="PCA"):
    return types.SimpleNamespace(vertPos_bases_type=kind, vertPos_numComponents=K, q_support=support,
                                 store_vertPos_PCA_sing_val=False, vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25,
                                 q_standarize=True, q_massWeight=False, q_orthogonal=False, vertPos_output_directory=".",
                                 name="bench", splocs_max_itrs=20, splocs_admm_num_itrs=10, splocs_lambda=2.0, splocs_rho=10.0,
                                 vertPos_rest_shape="first")


def _fixture_mesh(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return g["rest"], g["tris"].astype(np.int64), g


def other_config_pos(tag, fixture, support, kind, cpu=True):
    """configs 2 / 3: the real rest mesh (from the committed fixture), seeded synthetic frames of SURVEY.md 8(d)."""
    from animsnapbases_amd import posComponents, posSnapshots
    from oracle import asb_oracle as orc
    rest, tris, g = _fixture_mesh(fixture)
    F, K = int(g["F"]), int(g["param_vertPos_numComponents"])
    N = rest.shape[0]
    verts = orc.synth_snapshots(rest, F, rank=int(g["rank"]), noise=float(g["noise"]), seed=int(g["seed"]), kind=str(g["kind"]))
    holder = {}

    def prep():
        holder["snaps"] = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False)
        if support == "local":      # the geodesic set-up is lazy (global support never pays it): here it belongs to the preparation
            holder["snaps"].compute_geodesic_distance.prepare()
            holder["snaps"]._engine.sync()
    t_prep, _ = _timed(prep, lambda: None)
    snaps = holder["snaps"]
    comp = posComponents(_pos_param(K, support, kind), snaps)
    sync = snaps._engine.sync
    # every leg is timed on its SECOND call (the first pays for the context's allocations and code-object loads: 9 ms of the
    # SPLOCS leg's first call are two such holes); `cold_ms` keeps the first
    cold_ms, _ = _timed(comp.compute_components_store_singvalues, sync)       # warm-up
    if snaps.compute_geodesic_distance is not None:
        snaps.compute_geodesic_distance._cache.clear()                # the timed call computes its distance fields itself
    ms, _ = _timed(comp.compute_components_store_singvalues, sync)
    assert comp.selected_vertices.tolist() == g["idx"].tolist()       # sanity of what was timed (parity: tests/)
    c = 1 if support == "global" else 2
    alg = 24.0 * N * F * (1 + c * K)
    if kind == "SPLOCS":
        its, admm = 20, 10
        alg += its * (48.0 * N * F + 8.0 * 3 * N * K * (2 + 4 * admm))
    out = {"workload": "%s: %s (%d verts) x %d frames, %s K=%d support=%s%s" %
                       (tag, str(g["mesh"]), N, F, kind, K, support, ", 20 outer x 10 ADMM" if kind == "SPLOCS" else ""),
           "ms": ms, "cold_ms": cold_ms, "snapshots_per_s": F / (ms * 1e-3), "prepare_ms": t_prep,
           "roofline": {"bound": "hbm", "level": "call", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg,
                        "note": "SURVEY 8(d) algorithmic bytes of the whole call / its wall time; the %.0f MB tensor %s the 256 MB "
                                "Infinity Cache, so this is a nominal HBM figure" % (24.0 * N * F / 1e6, "fits" if 24.0 * N * F <= 256e6 else "exceeds")}}
    if cpu:
        cores, model = _cpu_info()
        geo = orc.Geodesics(verts[0], tris) if support == "local" else None
        pre = orc.prepare_snapshots(verts, "first", True)
        if kind == "PCA":
            med, ts = _median3(lambda: orc.extract_k_components(pre["snapTensor"], K, support, geo, 0.1, 0.25))
            out["cpu_baseline"] = dict(value=F / med, unit="snapshots/s", cores=cores, kind="port", cpu_model=model,
                                       runs_s=[round(t, 3) for t in ts], sample="NumPy oracle, the full workload, median of 3")
        else:
            ks = 4          # bounded: 4 of the 64 components, 1 of the 20 outer iterations; both loops are linear in K

            def sample():
                d = orc.extract_k_components(pre["snapTensor"], ks, support, geo, 0.1, 0.25)
                t1 = time.perf_counter()
                orc.splocs_glob_optimization(pre["snapTensor"], d["comps"], d["weigs"], d["R"], geo, 0.1, 0.25, 1, 10, 2.0, 10.0)
                sample.t_splocs.append(time.perf_counter() - t1)
            sample.t_splocs = []
            med, ts = _median3(sample)
            t_s = float(np.median(sample.t_splocs))
            full = (med - t_s) * (K / ks) + t_s * (K / ks) * 20
            out["cpu_baseline"] = dict(value=F / full, unit="snapshots/s", cores=cores, kind="port", cpu_model=model,
                                       runs_s=[round(t, 3) for t in ts],
                                       sample="NumPy oracle on %d of %d components and 1 of 20 SPLOCS outer iterations at full N, F "
                                              "(median of 3: %.2f s, of which SPLOCS %.2f s); scaled linearly in K and iterations"
                                              % (ks, K, med, t_s))
    return out


def structured_tensor(dev, kind, N, F, seed):
    """Synthetic STRUCTURED inputs of config 4's shape (F, N, 3), seeded, generated on the device ("parity on
    low-rank + noise"; what posComponents.py:67-129 is fed in practice is low-rank and localised):
      lowrank         rest + coef (F x 50) . modes (50 x 3N) + 1e-4 noise, modes ~ N(0, 0.02^2), coef[:, j] ~ N(0, 0.9^2j)
      slow_spectrum   the same with rank 200 and coef[:, j] ~ N(0, 0.97^2j)
      bumps           50 localised modes: Gaussian bumps (radius 0.1 - 0.35 of the unit cube the rest vertices fill) along random
                      directions, amplitude 0.1, coef as lowrank, + 1e-4 noise
      rank_deficient  rank 40 (as lowrank) + 1e-12 noise: K = 128 reaches far beyond the numerical rank"""
    import torch
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    f64 = dict(dtype=torch.float64, device=dev)
    if kind == "bumps":
        r = 50
        rest = torch.rand((N, 3), generator=gen, **f64)
        modes = torch.empty((r, N, 3), **f64)
        for j in range(r):
            c = rest[int(torch.randint(N, (1,), generator=gen, device=dev))]
            rad = 0.1 + 0.25 * float(torch.rand((1,), generator=gen, **f64))
            d = torch.randn((3,), generator=gen, **f64)
            modes[j] = torch.exp(-((rest - c) ** 2).sum(1) / rad ** 2)[:, None] * (d / d.norm())[None] * 0.1
        coef = torch.randn((F, r), generator=gen, **f64) * (0.9 ** torch.arange(r, **f64))[None]
        Xd = rest.reshape(1, -1) + coef @ modes.reshape(r, -1)
        noise = 1e-4
    else:
        r, decay, noise = {"lowrank": (50, 0.9, 1e-4), "slow_spectrum": (200, 0.97, 1e-4), "rank_deficient": (40, 0.9, 1e-12)}[kind]
        rest = torch.randn((N * 3,), generator=gen, **f64)
        coef = torch.randn((F, r), generator=gen, **f64) * (decay ** torch.arange(r, **f64))[None]
        modes = 0.02 * torch.randn((r, N * 3), generator=gen, **f64)
        Xd = rest[None] + coef @ modes
    Xd += noise * torch.randn((F, N * 3), generator=gen, **f64)
    del coef, modes, rest
    torch.cuda.synchronize()
    return Xd.reshape(F, N, 3)


STRUCTURED_NOTE = {
    "lowrank": "low rank (50) + 1e-4 noise",
    "slow_spectrum": "slowly decaying spectrum (rank 200, 0.97^j) + 1e-4 noise",
    "bumps": "50 localised modes (Gaussian bumps) + 1e-4 noise",
    "rank_deficient": "rank 40 + 1e-12 noise: K = 128 far beyond the numerical rank",
}


def other_config_c4_structured(dev, kind="lowrank", N=100000, F=2000, K=128, seed=77, residual_too=False):
    """config 4's shape on STRUCTURED data (structured_tensor): every strong component reshuffles all energies, so a read of X
    commits fewer components than on the random tensor of the headline.  residual_too: the same input through the residual loop
    (ASB_DEFLATE_MODE=residual: one read + one write of R per component, the reference's own algorithm) beside it."""
    import torch
    from animsnapbases_amd import posComponents, posSnapshots
    Xd = structured_tensor(dev, kind, N, F, seed)
    holder = {}

    def prep():
        holder["snaps"] = posSnapshots.from_device(Xd.data_ptr(), F, N, rest_shape="first", standarize=True, keepalive=Xd)
        holder["snaps"]._engine.sync()
    t_prep, _ = _timed(prep, torch.cuda.synchronize)
    snaps = holder["snaps"]
    comp = posComponents(_pos_param(K, "global"), snaps)
    sync = snaps._engine.sync
    _timed(lambda: comp.extract_k_components(None), sync)                 # warm-up
    ms, _ = _timed(lambda: comp.extract_k_components(None), sync)
    st = snaps._engine.deflate_stats()
    # (beyond the numerical rank a vertex may be taken again: it has three directions to give)
    assert kind == "rank_deficient" or len(set(comp.selected_vertices.tolist())) == K
    assert np.isfinite(comp.measures_at_largeDeforVerts).all()
    reads = st["panels"] + st.get("energy_passes", 0) + st.get("refreshes", 0)
    ksw = st.get("residual_switch_at", -1)
    # what THIS run moved: a read of X per panel pass / refresh, and one read + one write of R per component of the residual loop
    nbytes = reads * 24.0 * N * F + (48.0 * N * F * (K - ksw) + 24.0 * N * F if ksw >= 0 else 0.0)
    out = {"workload": "config4 shape, %s: %d verts x %d frames, PCA K=%d global" % (STRUCTURED_NOTE[kind], N, F, K),
           "ms": ms, "snapshots_per_s": F / (ms * 1e-3), "prepare_ms": t_prep, "reads_of_X": reads, "panels": st["panels"],
           "refreshes": st.get("refreshes", 0), "sketch_replays": st.get("sketch_runs", 0),
           "reads_with_predicted_candidates": st.get("sketch_reads", 0), "residual_switch_at": ksw,
           "roofline": {"bound": "hbm", "level": "call", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "(reads of X) x 24 N F bytes (+ 48 N F per component of the residual loop behind a switch) / wall "
                                "time: what THIS algorithm moves on this data"}}
    if residual_too:
        comp.deflate_mode = "residual"
        _timed(lambda: comp.extract_k_components(None), sync)
        ms_r, _ = _timed(lambda: comp.extract_k_components(None), sync)
        out["residual_mode_ms"] = ms_r
        out["note"] = "residual_mode_ms: the same input through the residual loop from the start (ASB_DEFLATE_MODE=residual)"
    del comp, snaps, holder
    return out


def other_config_c4_lowrank(dev, N=100000, F=2000, K=128, r=50, seed=77):
    return other_config_c4_structured(dev, "lowrank", N, F, K, seed)


def other_config_c4_lowrank_seeds(dev, seeds=(77, 78, 79)):
    """the low-rank leg on three tensors: the greedy sequence is chaotic in which read a rejection falls (6 or 7 reads per tensor,
    15-19 ms), so the leg reports the mean; `each` holds the single runs."""
    import torch
    runs = []
    for sd in seeds:
        runs.append(other_config_c4_lowrank(dev, seed=sd))
        torch.cuda.empty_cache()
    out = dict(runs[0])
    ms = float(np.mean([r["ms"] for r in runs]))
    reads = float(np.mean([r["reads_of_X"] for r in runs]))
    out.update(ms=ms, snapshots_per_s=runs[0]["snapshots_per_s"] * runs[0]["ms"] / ms, reads_of_X=reads, seeds=list(seeds),
               each=[{k: r[k] for k in ("ms", "reads_of_X", "sketch_replays", "reads_with_predicted_candidates")} for r in runs])
    out["roofline"] = dict(runs[0]["roofline"])
    scale = (reads / runs[0]["reads_of_X"]) * (runs[0]["ms"] / ms)
    out["roofline"]["achieved"] = runs[0]["roofline"]["achieved"] * scale
    out["roofline"]["frac"] = runs[0]["roofline"]["frac"] * scale
    out["workload"] += " (mean of %d tensors)" % len(seeds)
    return out


def other_config_c4_seeds(dev, seeds=(1, 2, 3, 4, 5, 6), N=100000, F=2000, K=128):
    """config 4 on OTHER random tensors than the one the headline is timed on.  How many of the greedy steps of a read stand
    depends on the draw: the first components remove the constant-in-time direction and leave random cross terms of about 1 % of the
    energies behind, the size of the spread among the leading few hundred vertices, so on most tensors the first read (candidates
    guessed from the energies without that direction) is cut short somewhere and the run takes three or four reads instead of two.
    The headline's tensor (seed 1234) is one of the two-read draws; this leg says what the others cost."""
    import torch
    from animsnapbases_amd import posComponents, posSnapshots
    ms, reads = [], []
    for sd in seeds:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(sd))
        Xd = torch.rand((F, N, 3), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        torch.cuda.synchronize()
        snaps = _quiet(lambda: posSnapshots.from_device(Xd.data_ptr(), F, N, rest_shape="first", standarize=True, keepalive=Xd))
        comp = posComponents(_pos_param(K, "global"), snaps)
        sync = snaps._engine.sync
        _timed(lambda: comp.extract_k_components(None), sync)                 # warm-up
        t, _ = _timed(lambda: comp.extract_k_components(None), sync)
        st = snaps._engine.deflate_stats()
        assert len(set(comp.selected_vertices.tolist())) == K
        ms.append(t)
        reads.append(st["panels"] + st.get("energy_passes", 0) + st.get("refreshes", 0))
        del comp, snaps, Xd
        torch.cuda.empty_cache()
    return {"workload": "config4 on %d other U[-1,1) tensors (torch seeds %s): %d verts x %d frames, PCA K=%d global" %
                        (len(seeds), list(seeds), N, F, K),
            "ms": float(np.mean(ms)), "ms_each": ms, "reads_of_X_each": reads, "snapshots_per_s": F / (float(np.mean(ms)) * 1e-3),
            "note": "mean over the seeds; the headline's tensor (seed 1234) needs 2 reads"}


def other_config_c4_forced(dev, N=100000, F=2000, K=128, steps=10, seed=1234):
    """The headline's workload through the MULTI-RANK protocol on one rank (Comm(force_collectives=True) over a single-rank RCCL
    group): every collective of a multi-GPU run really issued -- the start-up exchange, per read of X the two all-gathers
    (energies; candidate rows + ids) and the one min-all-reduce of the tile verdicts --, the candidates assembled from the gathered
    buffer, the read through asb_panel_read_run / _commit.  What the protocol costs on top of the fused single-rank driver."""
    import torch
    import torch.distributed as dist
    from animsnapbases_amd import Comm, posComponents, posSnapshots
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        comm = Comm(force_collectives=True)
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed))
        Xd = torch.rand((F, N, 3), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        torch.cuda.synchronize()
        snaps = _quiet(lambda: posSnapshots.from_device(Xd.data_ptr(), F, N, rest_shape="first", standarize=True, comm=comm, keepalive=Xd))
        comp = posComponents(_pos_param(K, "global"), snaps)
        eng = snaps._engine

        def sync():
            torch.cuda.synchronize()
            eng.sync()
        for _ in range(2):
            _timed(lambda: comp.extract_k_components(None), sync)
        t0 = time.perf_counter()
        for _ in range(steps):
            _quiet(lambda: comp.extract_k_components(None))
        sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        st = eng.deflate_stats()
        assert len(set(comp.selected_vertices.tolist())) == K
        out = {"workload": "config4 through the multi-rank protocol on ONE rank (single-rank RCCL group, every collective issued): "
                           "%d verts x %d frames, PCA K=%d global" % (N, F, K),
               "ms": ms, "snapshots_per_s": F / (ms * 1e-3), "steps": steps, "reads_of_X": st["panels"] + st.get("refreshes", 0),
               "panel_kernel_fallbacks": st.get("coop_fallbacks", 0),
               "note": "compare with the headline's ms_per_step (the fused single-rank driver on the same tensor)"}
        del comp, snaps, Xd
    finally:
        if own_group:
            dist.destroy_process_group()
    return out


def other_config_c5(dev, cpu=True, ep=50000, F=4000, K=256):
    """config 5: constraint-projection snapshots 50 000 x 3 rows x 4 000 frames, POD (pod_vectorized) K = 256 + DEIM."""
    import torch
    from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    r = 40
    coef = torch.randn((F, r), dtype=torch.float64, device=dev, generator=gen) * \
        (0.85 ** torch.arange(r, dtype=torch.float64, device=dev))[None]
    modes = torch.randn((r, ep * 3), dtype=torch.float64, device=dev, generator=gen)
    X0 = 0.1 + coef @ modes                                            # synthetic INPUT only (low rank + noise)
    X0 += 1e-5 * torch.randn((F, ep * 3), dtype=torch.float64, device=dev, generator=gen)
    del coef, modes
    torch.cuda.synchronize()
    param = types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=F, constProj_p_size=1,
                                  constProj_massWeight=False, constProj_standarize=True, constProj_orthogonal=False,
                                  constProj_basis_type="pod_vectorized", deim_desired_num_components=K,
                                  constProj_store_sing_val=False, constProj_output_directory=".", name="c5", constProj_name="v")
    # Two cycles on ONE engine, the second reported -- like every other leg (a warm-up call, then the timed one): the first
    # cycle of a context pays for its allocations (the DEIM loop: 35 ms warm, 70 - 120 ms cold), which `cold` records.
    from animsnapbases_amd import HipEngine
    eng = HipEngine(torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    cold = None
    for cyc in range(2):
        Xd = X0.clone()                                                # (the preparation standardises its input in place)
        torch.cuda.synchronize()
        ns = nonlinearSnapshots(param, frames_device=(Xd.data_ptr(), F, ep), keepalive=Xd, engine=eng)
        ns.config()
        t_prep, _ = _timed(ns.snapshots_prepare, torch.cuda.synchronize)
        del Xd
        sync = eng.sync
        cc = constraintsComponents(param, ns)
        cc.config()
        t_pod, _ = _timed(cc.compute_components_store_singvalues, sync)
        t_post, _ = _timed(cc.post_process_components, sync)
        t_deim, _ = _timed(cc.deim, sync)
        assert len(set(cc.geom_Pt.tolist())) == K
        if cyc == 0:
            cold = {"prepare_ms": t_prep, "pod_ms": t_pod, "post_process_ms": t_post, "deim_ms": t_deim}
            del cc, ns
    del X0
    M = 3 * ep
    flops = 2.0 * M * F * F + 2.0 * M * F * K
    # `ms` = POD + post-processing + DEIM (round 3 left the post-processing out of it)
    out = {"workload": "config5: %d x 3 constraint rows x %d frames, pod_vectorized K=%d + post-processing + DEIM, 1 GPU" % (ep, F, K),
           "ms": t_pod + t_post + t_deim, "snapshots_per_s": F / ((t_pod + t_post + t_deim) * 1e-3), "prepare_ms": t_prep,
           "pod_ms": t_pod, "post_process_ms": t_post, "deim_ms": t_deim, "cold": cold,
           "pod_levels": getattr(cc, "pod_levels", 1), "pod_power_steps": getattr(cc, "pod_power_steps", 0),
           "roofline": {"bound": "mfma", "level": "call", "achieved": flops / (t_pod * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flops / (t_pod * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops": flops,
                        "note": "SURVEY 8(d): 2 M F^2 + 2 M F K f64 flop / POD wall time against the FP64 matrix peak"}}
    del cc, ns, eng
    torch.cuda.empty_cache()
    if cpu:
        from oracle import asb_oracle as orc
        import scipy.linalg as sla
        cores, model = _cpu_info()
        rng = np.random.default_rng(5)
        ts_, ms_ = [], (6000, 12000)
        for m in ms_:                    # two bounded samples, linear fit t = a + b M (the F^3 part does not scale with M)
            A = rng.normal(size=(m, 40)) @ rng.normal(size=(40, F)) + 1e-5 * rng.normal(size=(m, F))
            med, _ = _median3(lambda: sla.svd(A, full_matrices=False))
            ts_.append(med)
        b = (ts_[1] - ts_[0]) / (ms_[1] - ms_[0])
        a = ts_[0] - b * ms_[0]
        full = a + b * M
        out["cpu_baseline"] = dict(value=F / full, unit="snapshots/s", cores=cores, kind="port", cpu_model=model,
                                   sample="scipy.linalg.svd (the oracle's pod_vectorized call) on %d x %d and %d x %d samples, "
                                          "median of 3 each (%.2f s, %.2f s); linear fit in the row count extrapolated to %d rows "
                                          "(POD only; DEIM not included)" % (ms_[0], F, ms_[1], F, ts_[0], ts_[1], M))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--verts", type=int, default=100000)
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--comps", type=int, default=128)
    ap.add_argument("--seed", type=int, default=1234, help="seed of the synthetic tensor of the timed steps (rank r uses seed + r)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--legs", default="", help="comma-separated subset of the other_configs legs to run (default: all)")
    ap.add_argument("--cpu-budget", type=float, default=18.0)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default): --verts vertices in all, sharded over the ranks (config 4 is one fixed job); "
                         "weak: --verts vertices PER RANK (the tensor grows with the GPUs)")
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
    # Launch rehearsal on a ONE-GPU box (tests/test_gpu_launch.py): ASB_BENCH_ONE_DEVICE=1 puts every rank on device 0 and
    # ASB_BENCH_BACKEND=gloo exchanges through the host (Comm's staged collectives) -- rank / seed / partition / barrier /
    # teardown / the one-JSON-line contract under the real launcher; the numbers of such a run mean nothing.
    one_device = os.environ.get("ASB_BENCH_ONE_DEVICE", "0") == "1"
    backend = os.environ.get("ASB_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # ASB_FORCE_COLLECTIVES=1 (with --gpus 1): single-rank RCCL group, multi-rank protocol -- measures what the
    # per-panel collectives and their host synchronisation cost on top of the kernels
    forced = os.environ.get("ASB_FORCE_COLLECTIVES", "0") == "1"
    if world > 1 or forced:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from animsnapbases_amd import Comm, partition, posComponents, posSnapshots

    F, N, K = args.frames, args.verts, args.comps
    if args.scaling == "weak":
        N = args.verts * world
    comm = Comm()
    v0, n_loc = partition(N, world)[rank]
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + rank)
    Xd = torch.rand((F, n_loc, 3), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    torch.cuda.synchronize()

    quiet = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(quiet):
        snaps = posSnapshots.from_device(Xd.data_ptr(), F, n_loc, rest_shape="first", standarize=True, comm=comm,
                                         keepalive=Xd)
    snaps._engine.sync()
    torch.cuda.synchronize()
    prepare_ms = (time.perf_counter() - t0) * 1e3           # layout change + rest shape + standardisation of the resident tensor
    del Xd
    torch.cuda.empty_cache()
    comp = posComponents(_pos_param(K, "global"), snaps)
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
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity of the result that was just timed (not a parity test: see tests/)
    assert len(set(comp.selected_vertices.tolist())) == K
    assert np.isfinite(comp.measures_at_largeDeforVerts).all()
    stats = eng.deflate_stats()
    # the co-resident panel kernel's exchange timing out (another process on the GPU, a rank that lost its CUs) is survived --
    # all ranks repeat the panel through the two-kernel loop together and stay on it -- but it must not pass unnoticed
    fallbacks = int(comm.allreduce_sum([float(stats.get("coop_fallbacks", 0))])[0]) if world > 1 else int(stats.get("coop_fallbacks", 0))
    if fallbacks and rank == 0:
        print("[bench] WARNING: the panel kernel's exchange timed out %d time(s) over the ranks: those panels were redone by the "
              "two-kernel loop and the timed steps include it (is something else running on these GPUs?)" % fallbacks, file=sys.stderr)
    # the (K, N, 3) basis back in host memory, as the reference leaves it (rank 0's share; pageable destination)
    t0 = time.perf_counter()
    basis = comp.comps if world == 1 else eng.results(want_comps=True, want_weigs=False)["comps"]
    download_ms = (time.perf_counter() - t0) * 1e3
    basis_bytes = basis.nbytes
    del basis

    # Steady-state end to end (one GPU): a SECOND animation on the same context -- its buffers exist, the basis streams into
    # the context's pinned host buffer while the second read of X still computes (HipEngine.components_stream), and `comps`
    # is a plain ndarray over that buffer.  Timed as ONE region, input tensor in HBM to basis in host memory, no
    # synchronisation in between; the first cycle warms the path up (pinned pages, allocations), the second is reported.
    steady = None
    if world == 1 and not forced:
        eng.components_stream(True)
        for cyc in range(2):
            gen.manual_seed(4321 + cyc)
            Xd2 = torch.rand((F, n_loc, 3), dtype=torch.float64, device=dev, generator=gen) * 2 - 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(quiet):
                snaps2 = posSnapshots.from_device(Xd2.data_ptr(), F, n_loc, rest_shape="first", standarize=True, comm=comm,
                                                  keepalive=Xd2, engine=eng)
                t1 = time.perf_counter()
                comp2 = posComponents(_pos_param(K, "global"), snaps2)
                comp2.extract_k_components(None)
                t2 = time.perf_counter()
                basis2 = comp2.comps
            t3 = time.perf_counter()
            assert isinstance(basis2, np.ndarray) and basis2.shape == (K, n_loc, 3) and np.isfinite(basis2[-1]).all()
            steady = {"prepare_ms": (t1 - t0) * 1e3, "step_ms": (t2 - t1) * 1e3, "basis_wait_ms": (t3 - t2) * 1e3,
                      "total_ms": (t3 - t0) * 1e3, "snapshots_per_s": F / (t3 - t0)}
            del basis2, comp2, snaps2, Xd2
        eng.components_stream(False)
        torch.cuda.empty_cache()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = F * args.steps / dt
        mode = "project" if getattr(eng, "mode", 0) == 1 else "residual"
        nsweep = 1
        # dominant kernel: one streaming pass over this rank's shard per launch; algorithmic bytes per launch = 24 n_loc F
        # (c = 1: one read of the shard)
        alg_bytes = 24.0 * n_loc * F / nsweep
        avg_ms = kern_ms / max(launches, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if launches else None
        # HBM traffic of that kernel from the PMC counters: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
        # this build at this shape (tools/summarise_pmc.py), an OFFLINE measurement committed under profiles/ -- counters
        # cannot be read from inside the run
        traffic, traffic_src = None, None
        # default settings: up to four 16-column sub-panels per read of X (k_project_l2d for 4, k_project_l2w for 2 - 3);
        # ASB_DOUBLE_PANELS=0 reads X once per 16-column panel (k_project_l2s)
        wide = (mode == "project" and os.environ.get("ASB_DOUBLE_PANELS", "1") != "0" and os.environ.get("ASB_WIDE_VARIANT", "4") == "4")
        kname = "k_project_l2s<4, 2, 2, 1>"
        try:
            pm = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
            if mode == "project" and (N, F, world) == (100000, 2000, 1) and os.environ.get("ASB_L2_VARIANT", "4") == "4":
                if wide:        # the multi-tile launches of a step: the average launch, as `achieved` is
                    ks = [v for k, v in pm["kernels"].items() if k.startswith("k_project_l2w<4, 1, 2, ") or k.startswith("k_project_l2d<")]
                    if ks:
                        traffic = sum(v["hbm_bytes"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks)
                elif kname in pm["kernels"]:
                    traffic = pm["kernels"][kname]["hbm_bytes"]
                if traffic is not None:
                    traffic_src = PMC_PROFILE + " (offline rocprofv3 --pmc passes of this build and shape, not this run)"
        except Exception:
            pass
        # step level: X is read once per panel pass (+ once for the initial energies when they are not carried over from
        # the standardisation sweep); 24 N F bytes each
        reads = launches / max(args.steps, 1) / nsweep + stats.get("energy_passes", 1)      # both per step
        step_bytes = reads * 24.0 * N * F
        step_gbs = step_bytes / (ms_step * 1e-3) / 1e9 / world
        # f64 MFMA work of the average launch: 2 * rows * F flops per component column, K columns per step
        flops_launch = (2.0 * 3 * n_loc * F * K * args.steps / launches) if (launches and mode == "project") else 0.0
        tflops = flops_launch / (avg_ms * 1e-3) / 1e12 if launches else None
        out = {
            "metric": "snapshots/sec (SVD+SPLOCS) for n_verts x n_frames; basis Frobenius err vs ref",
            "value": value, "unit": "snapshots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config4: synthetic U[-1,1) %d verts x %d frames, greedy-deflation PCA K=%d, "
                                   "global support, standardised, vertex rows sharded over %d GPU(s)" % (N, F, K, world),
                       "n_verts": N, "n_frames": F, "K": K,
                       "parallelism": "vertex-shard x%d" % world + (" (multi-rank protocol forced)" if forced else "") +
                                      (" (LAUNCH REHEARSAL: all ranks on one device, %s collectives through the host -- not a measurement)" % backend
                                       if (one_device or backend != "nccl") else "")},
            # The dominant kernel against BOTH of its ceilings, the binding one first.  One launch reads the shard once
            # (24 n F bytes) and does 2 * 3n * F flops per component column it projects on; at the peaks that is 0.60 ms
            # of HBM and 0.245 ms of f64 MFMA per 16 columns, so launches with 3 or 4 sub-panels (48 / 64 columns, the usual
            # case) are MFMA-bound and launches with 1 or 2 HBM-bound.  `bound` follows the average launch.
            "roofline": dict(
                         ([("bound", "mfma"), ("achieved", tflops), ("peak", FP64_MFMA_PEAK_TFLOPS), ("unit", "TFLOP/s"),
                           ("frac", tflops / FP64_MFMA_PEAK_TFLOPS),
                           ("hbm", {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS})]
                          if (launches and mode == "project" and flops_launch / (FP64_MFMA_PEAK_TFLOPS * 1e12) > alg_bytes / (HBM_PEAK_GBS * 1e9)) else
                          [("bound", "hbm"), ("achieved", achieved), ("peak", HBM_PEAK_GBS), ("unit", "GB/s"),
                           ("frac", (achieved / HBM_PEAK_GBS) if achieved else None),
                           ("mfma", ({"achieved": tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tflops / FP64_MFMA_PEAK_TFLOPS} if (launches and mode == "project") else None))]),
                         traffic=traffic, traffic_source=traffic_src,
                         kernel=(("k_project_l2d<4,3,1,3> / k_project_l2w<4,1,2,NCT,2,1> (f64-MFMA projection on 4 / NCT <= 3 sixteen-column "
                                  "sub-panels, 1 launch = one read of X per up to 64 components; config 4: two launches with 4 "
                                  "sub-panels)" if wide else
                                  "k_project_l2s<4,2,2,1> (f64-MFMA panel projection, one launch = one read of X per 16-column panel)")
                                 if mode == "project" else "k_stream<T,E2,UPDATE> (deflation pass, read+write of R)"),
                         columns_per_launch=(K * args.steps / launches * nsweep if launches else None),
                         algorithmic_flops_per_launch=flops_launch,
                         algorithm=mode, panels_per_step=stats["panels"], panel_kernel_fallbacks=fallbacks,
                         refreshes=stats["refreshes"], launches=launches,
                         avg_launch_ms=avg_ms, algorithmic_bytes_per_launch=alg_bytes,
                         # the whole step against the same peak: what THIS algorithm has to read (one read of X per
                         # panel pass [+ initial energies]) / ms_per_step, per GPU
                         step={"reads_of_X": reads, "bytes": step_bytes, "achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS,
                                  "dominant_kernel_share": (kern_ms / args.steps) / ms_step if launches else None,
                                  # and against the MFMA peak: the projections of a step are 2 * 3N * F * K flops whatever
                                  # the number of reads
                                  "flops": 2.0 * 3 * N * F * K, "mfma_achieved": 2.0 * 3 * N * F * K / (ms_step * 1e-3) / 1e12 / world,
                                  "mfma_frac": 2.0 * 3 * N * F * K / (ms_step * 1e-3) / 1e12 / world / FP64_MFMA_PEAK_TFLOPS},
                         # SURVEY.md 8(d) priced a step at 24 N F (1 + K) bytes (one read of X per component); the panel
                         # algorithm commits up to 16 components per read, so that figure is not a roofline for it
                         survey_step_bytes=24.0 * N * F * (1 + K)),
            # steady state (what a second animation on a warm context costs) first; `cold` = the first cycle of the process
            # (allocation of the 4.8 GB shard inside prepare, the basis into freshly allocated pageable memory)
            "end_to_end": dict(steady or {}, basis_bytes=basis_bytes,
                               note="input already in HBM (F,N,3) -> basis (K,N,3) as an ndarray in (pinned) host memory, one timed "
                                    "region without intermediate synchronisation, second cycle on a warm context: prepare = "
                                    "vertex-major layout change + rest shape + standardisation (the calls' host time: the kernels "
                                    "overlap with what follows), step = extract_k_components with the basis streamed to the host "
                                    "behind each read of X, basis_wait = what is left of the copy afterwards",
                               cold={"prepare_ms": prepare_ms, "step_ms": ms_step, "basis_download_ms": download_ms,
                                     "total_ms": prepare_ms + ms_step + download_ms,
                                     "snapshots_per_s": F / ((prepare_ms + ms_step + download_ms) * 1e-3),
                                     "note": "first cycle of the process: prepare includes the allocation of the shard, the "
                                             "download goes into pageable memory"}),
        }
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would wait for it)
            out["cpu_baseline"] = cpu_baseline_c4(F, N, K, args.cpu_budget)
    if world == 1 and not args.no_other_configs and (N, F, K) == (100000, 2000, 128):
        del comp, snaps, eng
        torch.cuda.empty_cache()
        cpu = not args.no_cpu_baseline
        oc = {}
        for tag, fn in (("c4_lowrank", lambda: other_config_c4_lowrank_seeds(dev)),
                        ("c4_bumps", lambda: other_config_c4_structured(dev, "bumps", seed=5)),
                        ("c4_slow_spectrum", lambda: other_config_c4_structured(dev, "slow_spectrum", seed=5)),
                        ("c4_rank_deficient", lambda: other_config_c4_structured(dev, "rank_deficient", seed=5, residual_too=True)),
                        ("c4_other_seeds", lambda: other_config_c4_seeds(dev)),
                        ("c4_forced_collectives", lambda: other_config_c4_forced(dev)),
                        ("c2", lambda: other_config_pos("config2", "c2_bunny_pca_global", "global", "PCA", cpu)),
                        ("c2_local", lambda: other_config_pos("config2 (local support)", "c2_bunny_pca_local", "local", "PCA", cpu)),
                        ("c3", lambda: other_config_pos("config3", "c3_armadillo_splocs", "local", "SPLOCS", cpu)),
                        ("c5", lambda: other_config_c5(dev, cpu))):
            if args.legs and tag not in args.legs.split(","):
                continue
            try:
                oc[tag] = fn()
            except Exception as e:                        # the headline line must still be printed
                oc[tag] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        out["other_configs"] = oc
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
