"""CPU, world_size = 2, gloo: the product's multi-rank host logic (posSnapshots /
posComponents + Comm) driven through the CPU test double of the device engine.
Checks: 2-rank result == oracle single-process result (index sequence exact), every rank
holds the full gathered basis, uneven shards (odd N)."""
import os
import socket
import sys
import types

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, support, tmpdir, mode=None, device_select=True):
    import contextlib
    import io

    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.chdir(tmpdir)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from animsnapbases_amd import Comm, posComponents, posSnapshots
        from fake_engine import FakeEngine
        from oracle import asb_oracle as orc
        from conftest import align_signs, relerr

        rest, tris = orc.synth_mesh(7, 9, seed=3)               # N = 65 (odd: uneven shards)
        verts = orc.synth_snapshots(rest, 21, rank=5, seed=3, kind="bumps" if support == "local" else "iid")
        K = 5
        param = types.SimpleNamespace(vertPos_bases_type="PCA", vertPos_numComponents=K, q_support=support,
                                      store_vertPos_PCA_sing_val=True, vertPos_smooth_min_dist=0.1,
                                      vertPos_smooth_max_dist=0.35, q_standarize=True, q_massWeight=False,
                                      q_orthogonal=False, vertPos_output_directory=tmpdir, name="mr%d" % rank)
        comm = Comm()
        assert comm.world == world and comm.rank == rank
        FakeEngine.GLOBAL_TAU_ON_DEVICE = device_select
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False,
                                             engine=FakeEngine(), comm=comm)
            comp = posComponents(param, snaps)
            comp.deflate_mode = mode
            comp.compute_components_store_singvalues()
        if mode == "project":
            st = snaps._engine.deflate_stats_project()
            assert st["panels"] >= 2, st           # several panels really happened (tiny candidate capacity)
            # ... and some of them went on with unproven steps that the all-reduced check then kept or cut
            assert st["unproven_tried"] > 0 and 0 <= st["unproven_kept"] <= st["unproven_tried"], st
        assert snaps._engine.n_loc == comm.my_shard(65)[1] and sum(n for _, n in comm.shards(65)) == 65
        pre = orc.prepare_snapshots(verts, "first", True)
        assert abs(snaps.pre_scale_factor - pre["pre_scale_factor"]) < 1e-12 * pre["pre_scale_factor"]
        assert relerr(snaps.mean, pre["mean"]) < 1e-14
        assert relerr(snaps.snapTensor, pre["snapTensor"]) < 1e-13
        geo = orc.Geodesics(verts[0], tris) if support == "local" else None
        ref = orc.extract_k_components(pre["snapTensor"], K, support, geo, 0.1, 0.35)
        assert comp.selected_vertices.tolist() == ref["idx"].tolist()
        comps, weigs = comp.comps, comp.weigs
        assert comps.shape == (K, 65, 3)
        if support == "global":
            comps, weigs = align_signs(comps, weigs, ref["comps"])
        assert relerr(comps, ref["comps"]) < 1e-9
        assert relerr(weigs, ref["weigs"]) < 1e-9
        assert relerr(comp.measures_at_largeDeforVerts, ref["measures"]) < (1e-7 if mode == "project" else 1e-9)
        with contextlib.redirect_stdout(io.StringIO()):
            comp.post_process_components()
        post = orc.post_process_components(comp.comps * 0 + comps if False else ref["comps"], pre["pre_scale_factor"], pre["mean"])
        if support == "local":
            assert relerr(comp.comps, post) < 1e-9
        if rank == 0:
            assert os.path.exists(os.path.join(tmpdir, "mr0_posBases_pcaExtraction_singValues_errorNorm.csv"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("support", ["global", "local"])
def test_two_rank_gloo_matches_oracle(support, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, support, str(tmp_path), "residual"), nprocs=2, join=True)


@pytest.mark.parametrize("device_select", [True, False])
def test_two_rank_gloo_projection_mode(tmp_path, device_select):
    """The panel (projection-mode) multi-rank protocol: local thresholds, one all-gather of exported energies, global
    threshold (the engine's selection, or the torch fallback the driver uses when the table exceeds the selection
    kernel's LDS), padded candidate all-gather, replicated greedy steps, local projection -- through the CPU test double."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, "global", str(tmp_path), "project", device_select), nprocs=2, join=True)


def test_four_rank_gloo_projection_mode(tmp_path):
    """The same protocol over four processes (uneven shards of 65 vertices: 17, 16, 16, 16): per-rank counts, the packed
    rows + ids exchange and the min over ranks of the unproven-step checks with more than two participants."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(4, port, "global", str(tmp_path), "project", True), nprocs=4, join=True)


def test_unproven_steps_protocol_keeps_the_sequence(monkeypatch, tmp_path):
    """The panel driver's unproven steps (run_spec -> project_spec -> min over ranks -> commit) on the CPU test double, one
    process: uniform noise with rest_shape='first' makes the provable panels short; the checked extra steps must save
    panels and leave the oracle's vertex sequence untouched."""
    import contextlib
    import io

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from animsnapbases_amd import Comm, posComponents, posSnapshots
    from fake_engine import FakeEngine
    from oracle import asb_oracle as orc
    from conftest import align_signs, relerr

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    F, N, K = 48, 1500, 24
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    param = types.SimpleNamespace(vertPos_bases_type="PCA", vertPos_numComponents=K, q_support="global",
                                  store_vertPos_PCA_sing_val=False, vertPos_smooth_min_dist=0.1,
                                  vertPos_smooth_max_dist=0.35, q_standarize=True, q_massWeight=False,
                                  q_orthogonal=False, vertPos_output_directory=".", name="spec")
    pre = orc.prepare_snapshots(verts, "first", True)
    d = orc.extract_k_components(pre["snapTensor"], K, "global", None, 0.1, 0.4)
    stats = {}
    for spec in ("1", "0"):
        monkeypatch.setenv("ASB_SPEC_PANELS", spec)
        with contextlib.redirect_stdout(io.StringIO()):
            snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False, engine=FakeEngine(),
                                             comm=Comm(force_single=True))
            comp = posComponents(param, snaps)
            comp.deflate_mode = "project"
            comp._stepwise_panels = True
            comp.compute_components_store_singvalues()
        assert comp.selected_vertices.tolist() == d["idx"].tolist()
        comps, weigs = align_signs(comp.comps, comp.weigs, d["comps"])
        assert relerr(comps, d["comps"]) < 1e-9 and relerr(weigs, d["weigs"]) < 1e-9
        stats[spec] = snaps._engine.deflate_stats_project()
    assert stats["0"]["unproven_tried"] == 0
    assert stats["1"]["unproven_kept"] > 0 and stats["1"]["panels"] < stats["0"]["panels"], stats


def _comm_worker(rank, world, port):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from animsnapbases_amd import Comm, partition
        comm = Comm()
        for rep in range(2):            # the second round reuses the cached exchange tensors
            a = np.arange(5, dtype=np.float64) * (rank + 1) + rep
            assert np.array_equal(comm.allreduce_sum(a), np.arange(5) * sum(range(1, world + 1)) + world * rep)
            assert np.array_equal(comm.allreduce_max(a), np.arange(5) * world + rep)
            assert comm.allreduce_sum(3.5).shape == (1,)
            got = comm.all_gather_ints([rank, 10 * rank + rep, -7])
            assert got.shape == (world, 3) and got.dtype == np.int64
            assert got.tolist() == [[r, 10 * r + rep, -7] for r in range(world)]
        # arg-max over ranks: larger value wins, equal values -> the lowest index (NumPy's first maximum)
        val = np.array([1.0 + rank, 5.0, 2.0 - rank, 0.0])
        idx = np.array([100 + rank, 50 - rank, 7 + rank, 9 * (world - rank)], dtype=np.int64)
        best = comm.global_argmax(idx, val)
        assert best.tolist() == [100 + world - 1, 50 - (world - 1), 7, 9], best
        # uneven blocks along either axis come back in rank order on every rank
        N = 4 * world + 1
        v0, n = partition(N, world)[rank]
        full = np.arange(3 * N * 2, dtype=np.float64).reshape(3, N, 2)
        assert np.array_equal(comm.all_gather_rows(full[:, v0:v0 + n], N, axis=1), full)
        assert np.array_equal(comm.all_gather_rows(full[0, v0:v0 + n], N, axis=0), full[0])
        assert len(comm._bufs) <= 8              # exchange tensors are cached, not rebuilt per call
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_comm_collectives_gloo(world):
    """The small-value collectives of Comm (cached exchange tensors, all_reduce / all_gather_into_tensor only)."""
    import torch.multiprocessing as mp
    mp.spawn(_comm_worker, args=(world, _free_port()), nprocs=world, join=True)
