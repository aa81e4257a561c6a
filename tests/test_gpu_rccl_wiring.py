"""GPU (-m gpu): the torch.distributed/RCCL wiring of the multi-rank code path on a ONE-GPU box.

RCCL refuses two ranks on one device, so the process group here has a single rank -- but
``Comm(force_collectives=True)`` makes the product take its multi-rank branch and really issue
every collective (all-reduce / all-gather on device tensors through ProcessGroupNCCL) between
the engine's kernels.  What this covers that the other tests cannot: dtype / device / shape of
each exchanged tensor as RCCL sees it, and the ordering of RCCL's stream against the engine's
stream (the engine runs on torch's current stream).  Kernels with world > 1 are covered by
thread_comm (test_gpu_parity), the host protocol by the gloo tests.
"""
import contextlib
import io
import socket
import types

import numpy as np
import pytest

from conftest import align_signs, relerr
from oracle import asb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_comm():
    import torch
    import torch.distributed as dist
    from animsnapbases_amd import Comm
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    comm = Comm(force_collectives=True)
    assert comm.multi and comm.world == 1
    yield comm
    dist.destroy_process_group()


def _param(**over):
    base = dict(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                q_support="global", vertPos_numComponents=4, store_vertPos_PCA_sing_val=False,
                vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.35, vertPos_rest_shape="first",
                name="t", vertPos_output_directory=".")
    base.update(over)
    return types.SimpleNamespace(**base)


@pytest.mark.parametrize("mode,support", [("project", "global"), ("residual", "global"), (None, "local")])
def test_deflation_through_rccl(rccl_comm, mode, support):
    from animsnapbases_amd import posComponents, posSnapshots
    if support == "local":
        rest_v, tris = orc.synth_mesh(12, 17, seed=4)
        verts = orc.synth_snapshots(rest_v, 48, rank=6, seed=4, kind="bumps")
        K = 6
    else:
        verts, tris, K = np.random.default_rng(5).uniform(-1, 1, size=(64, 5003, 3)), None, 24
    param = _param(vertPos_numComponents=K, q_support=support, q_orthogonal=(support == "local"))
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False, comm=rccl_comm)
        assert snaps._engine.device_exchange          # records / candidates are exchanged as device tensors
        comp = posComponents(param, snaps)
        comp.deflate_mode = mode
        comp.compute_components_store_singvalues()
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris) if support == "local" else None
    ref = orc.extract_k_components(pre["snapTensor"], K, support, geo, 0.1, 0.35)
    assert comp.selected_vertices.tolist() == ref["idx"].tolist()
    comps, weigs = comp.comps, comp.weigs
    if support == "global":
        comps, weigs = align_signs(comps, weigs, ref["comps"])
    assert relerr(comps, ref["comps"]) < 1e-8
    assert relerr(weigs, ref["weigs"]) < 1e-8
    assert relerr(comp.measures_at_largeDeforVerts, ref["measures"]) < 1e-7
    if support == "local":                           # orth: Gram all-reduce + refinement all-reduce
        with contextlib.redirect_stdout(io.StringIO()):
            comp.post_process_components()
        want = orc.post_process_components(ref["comps"], pre["pre_scale_factor"], pre["mean"], orthogonal=True)
        got = comp.comps.copy()
        for k in range(K):
            for l in range(3):
                if np.dot(got[k, :, l], want[k, :, l]) < 0:
                    got[k, :, l] *= -1
        assert relerr(got, want) < 1e-7


def test_from_device_shard_through_rccl(rccl_comm):
    """The bench's construction: adopt a torch tensor in HBM as this rank's shard."""
    import torch
    from animsnapbases_amd import posComponents, posSnapshots
    F, N, K = 96, 7001, 20
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    Xd = torch.rand((F, N, 3), dtype=torch.float64, device="cuda", generator=gen) * 2 - 1
    verts = Xd.cpu().numpy()
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_device(Xd.data_ptr(), F, N, "first", True, comm=rccl_comm, keepalive=Xd)
        comp = posComponents(_param(vertPos_numComponents=K), snaps)
        comp.compute_components_store_singvalues()
    pre = orc.prepare_snapshots(verts, "first", True)
    ref = orc.extract_k_components(pre["snapTensor"], K, "global", None, 0.1, 0.35)
    assert comp.selected_vertices.tolist() == ref["idx"].tolist()
    comps, weigs = align_signs(comp.comps, comp.weigs, ref["comps"])
    assert relerr(comps, ref["comps"]) < 1e-8 and relerr(weigs, ref["weigs"]) < 1e-8


def test_splocs_through_rccl(rccl_comm):
    from animsnapbases_amd import posComponents, posSnapshots
    rest_v, tris = orc.synth_mesh(12, 17, seed=9)
    verts = orc.synth_snapshots(rest_v, 40, rank=6, seed=9, kind="bumps")
    K = 6
    param = _param(vertPos_numComponents=K, q_support="local", vertPos_bases_type="SPLOCS", vertPos_smooth_max_dist=0.4,
                   splocs_max_itrs=3, splocs_admm_num_itrs=4, splocs_lambda=2.0, splocs_rho=10.0)
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False, comm=rccl_comm)
        comp = posComponents(param, snaps)
        comp.compute_components_store_singvalues()
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], tris)
    d = orc.extract_k_components(pre["snapTensor"], K, "local", geo, 0.1, 0.4)
    s = orc.splocs_glob_optimization(pre["snapTensor"], d["comps"], d["weigs"], d["R"], geo, 0.1, 0.4, 3, 4, 2.0, 10.0)
    assert comp.splocs_centres.tolist() == s["idx"].tolist()
    assert np.allclose(comp.splocs_trace, s["trace"], rtol=1e-8)
    assert relerr(comp.splocs_comps, s["C"]) < 1e-8


def test_pod_deim_through_rccl(rccl_comm, tmp_path):
    """Config-5 path: F x F Gram all-reduce, back-projection, CholeskyQR2 Gram all-reduces, DEIM global arg-max."""
    from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
    rng = np.random.default_rng(23)
    ep, F, K = 1501, 64, 12
    frames = 0.1 + np.tensordot(rng.normal(size=(F, 20)) * (0.7 ** np.arange(20))[None], rng.normal(size=(20, ep, 3)), (1, 0)) \
        + 1e-6 * rng.normal(size=(F, ep, 3))
    param = types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=1,
                                  constProj_massWeight=False, constProj_standarize=True, constProj_orthogonal=True,
                                  constProj_basis_type="pod_vectorized", deim_desired_num_components=K,
                                  constProj_store_sing_val=False, constProj_output_directory=str(tmp_path), name="c5",
                                  constProj_name="verts")
    with contextlib.redirect_stdout(io.StringIO()):
        ns = nonlinearSnapshots(param, frames=frames, comm=rccl_comm)
        ns.config()
        ns.snapshots_prepare()
        cc = constraintsComponents(param, ns)
        cc.config()
        cc.compute_components_store_singvalues()
        cc.post_process_components()
        cc.deim()
    pre = orc.prepare_nonlinear_snapshots(frames, "first", True)
    pod = orc.pod_vectorized(pre["snapTensor"], K)
    assert relerr(cc.singular_values[:K], pod["S"][:K]) < 1e-10
    for l in range(3):
        assert np.allclose(cc.comps[:, :, l] @ cc.comps[:, :, l].T, np.eye(K), atol=1e-10)
    assert len(set(cc.geom_Pt.tolist())) == K


def test_config4_full_size_vs_reference_through_rccl(rccl_comm, tmp_path):
    """The headline configuration through the MULTI-RANK protocol (every collective issued, candidates assembled, sub-panels
    through asb_panel_sub_*) against the unmodified reference's run on the same input (tests/golden/c4_*.npz)."""
    from animsnapbases_amd import posComponents, posSnapshots
    from conftest import load_golden
    from config_fixtures import c4_frames, make_param
    from test_gpu_configs import check_config4
    g = load_golden("c4_uniform_pca_global")
    verts = c4_frames(g)
    param = make_param(g, vertPos_output_directory=str(tmp_path))
    snaps = posSnapshots.from_arrays(verts, None, param.vertPos_rest_shape, standarize=param.q_standarize,
                                     massWeight=param.q_massWeight, comm=rccl_comm)
    del verts
    comp = posComponents(param, snaps)
    with contextlib.redirect_stdout(io.StringIO()):
        comp.compute_components_store_singvalues()
    check_config4(g, snaps, comp, param, tmp_path)
