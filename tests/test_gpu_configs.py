"""GPU (-m gpu): BASELINE.json configs 2 and 3 AT SIZE -- the real bunny / armadillo rest meshes, the seeded
synthetic frames of SURVEY.md 8(d), K = 32 / K = 64 + SPLOCS 20 x 10 -- against compact fixtures produced by the
UNMODIFIED reference (oracle/gen_golden_configs.py): selected-vertex sequence bit-exact; measures, CSV,
pre_scale_factor, the basis at a seeded sample of vertices, its norms and seeded random projections of the basis and
the weights; for SPLOCS the printed trace, the centre sequence of all 20 outer iterations and the refined C, W.
"""
import numpy as np
import pytest

from conftest import load_golden, relerr
from config_fixtures import c4_frames, c5_frames, c5_probes, check_deflation, make_param, probes, regen_frames

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _run(g, tmp_path, mode=None, **over):
    from animsnapbases_amd import posComponents, posSnapshots
    verts = regen_frames(g)
    param = make_param(g, vertPos_output_directory=str(tmp_path), **over)
    snaps = posSnapshots.from_arrays(verts, g["tris"].astype(np.int64), param.vertPos_rest_shape,
                                     standarize=param.q_standarize, massWeight=param.q_massWeight)
    comp = posComponents(param, snaps)
    comp.deflate_mode = mode
    comp.compute_components_store_singvalues()
    return snaps, comp, param


@pytest.mark.parametrize("mode", [None, "residual", "project"])
def test_config2_bunny_pca_global(mode, tmp_path):
    """Config 2: bunny.obj (14 290 vertices) x 200 frames, PCA K = 32, global support; ``mode`` None is what a user
    gets (the 68 MB shard takes the cache-resident residual loop), the other two force each device algorithm."""
    g = load_golden("c2_bunny_pca_global")
    snaps, comp, param = _run(g, tmp_path, mode)
    check_deflation(g, snaps.pre_scale_factor, snaps.mean, comp.selected_vertices, comp.comps, comp.weigs,
                    comp.measures_at_largeDeforVerts, signed=False, tol=TOL,
                    mtol=TOL if mode != "project" else 1e-7,
                    csv=open(str(tmp_path / (param.name + "_posBases_pcaExtraction_singValues_errorNorm.csv"))).read())
    # post-processing (un-scale, + mean): a sign flip before "+ mean" is undone in the standardised space
    G, H, sv = probes(g)
    K = comp.numComp
    sign = np.sign(np.einsum("kp,kp->k", comp.comps.reshape(K, -1) @ G, g["comps_proj"]))
    comp.post_process_components()
    got = (comp.comps - snaps.mean[None]) * sign[:, None, None] + snaps.mean[None]
    assert relerr(got.reshape(K, -1) @ G, g["post_proj"]) < TOL
    assert relerr(got[:, sv, :], g["post_sample"]) < TOL
    assert relerr(np.sqrt((got.reshape(K, -1) ** 2).sum(1)), g["post_norms"]) < TOL


def test_config2_bunny_pca_local(tmp_path):
    """Config 2 with ``support='local'``: every step needs the heat-method distance field of the picked bunny vertex
    (dense device inverse here, SuperLU in the reference)."""
    g = load_golden("c2_bunny_pca_local")
    snaps, comp, param = _run(g, tmp_path)
    check_deflation(g, snaps.pre_scale_factor, snaps.mean, comp.selected_vertices, comp.comps, comp.weigs,
                    comp.measures_at_largeDeforVerts, signed=True, tol=1e-8, mtol=1e-8,
                    csv=open(str(tmp_path / (param.name + "_posBases_pcaExtraction_singValues_errorNorm.csv"))).read())
    assert (comp.weigs >= 0).all() and np.allclose(comp.weigs.max(axis=0), 1.0)
    G, H, sv = probes(g)
    K = comp.numComp
    comp.post_process_components()
    assert relerr(comp.comps.reshape(K, -1) @ G, g["post_proj"]) < 1e-8
    assert relerr(comp.comps[:, sv, :], g["post_sample"]) < 1e-8


def test_config3_armadillo_splocs(tmp_path, capsys):
    """Config 3: armadillo.obj (14 793 vertices) x 1000 frames, local-support deflation K = 64 followed by SPLOCS
    (20 outer x 10 ADMM iterations, lambda 2, rho 10) -- trace, centres and refined C / W of the reference."""
    g = load_golden("c3_armadillo_splocs")
    snaps, comp, param = _run(g, tmp_path)
    # comps / weigs are those of the local-support deflation, untouched by SPLOCS
    check_deflation(g, snaps.pre_scale_factor, snaps.mean, comp.selected_vertices, comp.comps, comp.weigs,
                    comp.measures_at_largeDeforVerts, signed=True, tol=1e-8, mtol=1e-8,
                    csv=open(str(tmp_path / (param.name + "_posBases_pcaExtraction_singValues_errorNorm.csv"))).read())
    itrs = int(g["param_splocs_max_itrs"])
    assert comp.splocs_trace.shape == (itrs, 2)
    assert comp.splocs_centres.tolist() == g["splocs_centres"].tolist()      # 20 x 64 centre vertices, bit-exact
    # the reference prints %f (6 decimals): compare parsed floats
    assert np.allclose(comp.splocs_trace, g["splocs_trace"], rtol=1e-8, atol=2e-6)
    G, H, sv = probes(g)
    K = comp.numComp
    C, W = comp.splocs_comps, comp.splocs_weigs
    assert relerr(C.reshape(K, -1) @ G, g["splocs_C_proj"][-1]) < 1e-7
    assert relerr(C[:, sv, :], g["splocs_C_sample"][-1]) < 1e-7
    assert relerr(np.sqrt((C.reshape(K, -1) ** 2).sum(1)), g["splocs_C_norms"][-1]) < 1e-7
    assert relerr(W.T @ W, g["splocs_WtW_last"]) < 1e-7
    if np.isfinite(g["splocs_W_proj"][-1]).all():
        assert relerr(H @ W, g["splocs_W_proj"][-1]) < 1e-7
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("itr ")]
    assert len(lines) == itrs and lines[0].startswith("itr 000, Energy =")


def check_config4(g, snaps, comp, param, tmp_path):
    """The whole of config 4 against the UNMODIFIED reference's run on the same 4.8 GB input (23 minutes of NumPy in the build
    container, oracle/gen_golden_configs.py c4): the 128 selected vertices bit-exact, sigma / residual norms / CSV,
    pre_scale_factor, the basis at 48 sampled vertices, its norms, 32 seeded projections of basis and weights, and the
    post-processed basis."""
    check_deflation(g, snaps.pre_scale_factor, snaps.mean, comp.selected_vertices, comp.comps, comp.weigs,
                    comp.measures_at_largeDeforVerts, signed=False, tol=TOL, mtol=1e-7,
                    csv=open(str(tmp_path / (param.name + "_posBases_pcaExtraction_singValues_errorNorm.csv"))).read())
    G, H, sv = probes(g)
    K = comp.numComp
    sign = np.sign(np.einsum("kp,kp->k", comp.comps.reshape(K, -1) @ G, g["comps_proj"]))
    comp.post_process_components()
    got = (comp.comps - snaps.mean[None]) * sign[:, None, None] + snaps.mean[None]
    assert relerr(got.reshape(K, -1) @ G, g["post_proj"]) < TOL
    assert relerr(got[:, sv, :], g["post_sample"]) < TOL
    assert relerr(np.sqrt((got.reshape(K, -1) ** 2).sum(1)), g["post_norms"]) < TOL


def test_config4_full_size_vs_reference(tmp_path):
    """BASELINE config 4 (the headline): U[-1,1) 100 000 vertices x 2 000 frames, K = 128, global support, standardised --
    the default device path (guessed first panel, four sub-panels per read, k_panel_multi, k_project_l2d)."""
    from animsnapbases_amd import posComponents, posSnapshots
    g = load_golden("c4_uniform_pca_global")
    verts = c4_frames(g)
    param = make_param(g, vertPos_output_directory=str(tmp_path))
    snaps = posSnapshots.from_arrays(verts, None, param.vertPos_rest_shape, standarize=param.q_standarize,
                                     massWeight=param.q_massWeight)
    del verts
    comp = posComponents(param, snaps)
    comp.compute_components_store_singvalues()
    st = snaps._engine.deflate_stats()
    assert st["panels"] <= 3                      # reads of X (2 on this input: 64 + 64 components)
    check_config4(g, snaps, comp, param, tmp_path)


def test_config5_full_size_vs_reference(tmp_path):
    """BASELINE config 5 at size: 50 000 x 3 rows x 4 000 frames, pod_vectorized K = 256 + post-processing + DEIM against
    the unmodified reference (SciPy gesdd on the 150 000 x 4 000 matrix + the lstsq loop): all 4 000 singular values that
    the data determine, all 256 vectors (sign-aligned), the post-processed basis and the whole 256-point DEIM sequence."""
    from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
    import types
    g = load_golden("c5_constraints_pod_deim")
    frames = c5_frames(g)
    K = int(g["K"])
    param = types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=1,
                                  constProj_massWeight=False, constProj_standarize=True, constProj_orthogonal=False,
                                  constProj_basis_type="pod_vectorized", deim_desired_num_components=K,
                                  constProj_store_sing_val=True, constProj_output_directory=str(tmp_path), name="c5",
                                  constProj_name="verts")
    ns = nonlinearSnapshots(param, frames=frames)
    del frames
    ns.config()
    ns.snapshots_prepare()
    assert abs(ns.pre_scale_factor - float(g["pre_scale_factor"])) < 1e-12 * float(g["pre_scale_factor"])
    Gp, sv = c5_probes(g)
    assert relerr(ns.mean.reshape(-1) @ Gp, g["mean_proj"]) < 1e-12
    cc = constraintsComponents(param, ns)
    cc.config()
    cc.compute_components_store_singvalues()
    S = cc.singular_values
    big = g["S"] > 1e-6 * g["S"][0]               # (the 320 that are signal; the rest is the 1e-9 noise floor)
    assert big.sum() >= K and relerr(S[big], g["S"][big]) < 1e-9
    comps = cc.comps.copy()
    P = comps.reshape(K, -1) @ Gp
    sign = np.sign(np.einsum("kp,kp->k", P, g["comps_proj"]))
    # accuracy: the Gram route + Rayleigh-Ritz alone leaves eps (sigma_0 / sigma_k)^2 OUTSIDE the K + 32 Ritz vectors -- 1e-6 to
    # 5e-6 on the last vectors here (sigma_256 = 3e-6 sigma_0; round 3's figure, ASB_POD_POWER=0) --; with the step of subspace
    # iteration of round 4 (asb_pod_power) every one of the 256 vectors is held to 1e-7, the strong ones to 1e-8
    per = np.sqrt((((P * sign[:, None]) - g["comps_proj"]) ** 2).sum(1) / (g["comps_proj"] ** 2).sum(1))
    print("config 5: per-vector error of the 32 projections: head", per[:4], "k=128", per[126:130], "tail", per[-4:],
          "largest %.2e at k = %d; power steps: %s" % (per.max(), int(per.argmax()), getattr(cc, "pod_power_steps", 0)))
    strong = g["S"][:K] > 1e-4 * g["S"][0]
    assert strong.sum() >= 128 and per[strong].max() < 1e-8
    assert per.max() < 1e-7
    assert relerr(P * sign[:, None], g["comps_proj"]) < 1e-7
    assert relerr(comps[:, sv, :] * sign[:, None, None], g["comps_sample"]) < 1e-7
    assert np.abs(comps.reshape(K, -1) @ comps.reshape(K, -1).T - np.eye(K)).max() < 1e-10
    # install the reference's signs (LAPACK's are arbitrary and "+ mean" is not symmetric), post-process, DEIM
    cc.comps = comps * sign[:, None, None]
    del comps
    cc.post_process_components()
    post = cc.comps
    assert relerr(post.reshape(K, -1) @ Gp, g["post_proj"]) < 1e-5
    assert relerr(post[:, sv, :], g["post_sample"]) < 1e-5
    assert relerr(np.sqrt((post.reshape(K, -1) ** 2).sum(1)), g["post_norms"]) < 1e-9
    cc.deim()
    same = np.asarray(cc.geom_Pt) == g["Pt"]
    print("config 5: DEIM points equal to the reference's:", int(same.sum()), "of", K, "first difference at",
          int(np.argmin(same)) if not same.all() else None)
    assert cc.geom_Pt.tolist() == g["Pt"].tolist()                       # 256 interpolation points, bit-exact
    assert cc.geom_alpha.tolist() == g["alpha"].tolist()
    assert cc.geom_alpha_ranges.tolist() == g["alpha_ranges"].tolist()
