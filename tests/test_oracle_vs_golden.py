"""CPU: the NumPy oracle against vectors produced by the UNMODIFIED reference
(oracle/gen_golden.py).  Index sequences equal, values <= 1e-12 relative."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import asb_oracle as orc

TOL = 1e-12
POS_CASES = ["pca_global_small", "pca_global_avg_mass_orth", "pca_global_nostd", "pca_global_medium",
             "pca_local_small", "splocs_small"]


def _prep(g):
    massL = None
    if bool(g["param_q_massWeight"]):
        mass, massL, inv = orc.factorize_masses(g["mass"])
        assert relerr(massL, g["massL"]) < TOL and relerr(inv, g["invMassL"]) < TOL
    return orc.prepare_snapshots(g["verts"], str(g["param_vertPos_rest_shape"]),
                                 bool(g["param_q_standarize"]), massL)


@pytest.mark.parametrize("name", POS_CASES)
def test_prepare_and_deflation(name):
    g = load_golden(name)
    pre = _prep(g)
    assert relerr(pre["mean"], g["mean"]) < TOL
    assert abs(pre["pre_scale_factor"] - float(g["pre_scale_factor"])) <= TOL * abs(float(g["pre_scale_factor"]))
    if "snapTensor" in g:
        assert relerr(pre["snapTensor"], g["snapTensor"]) < TOL
    support = str(g["param_q_support"])
    geo = orc.Geodesics(g["verts"][0] if str(g["param_vertPos_rest_shape"]) == "first" else g["verts"].mean(axis=0),
                        g["tris"]) if support == "local" else None
    K = int(g["param_vertPos_numComponents"])
    out = orc.extract_k_components(pre["snapTensor"], K, support, geo,
                                   float(g["param_vertPos_smooth_min_dist"]), float(g["param_vertPos_smooth_max_dist"]))
    assert out["idx"].tolist() == g["idx"].tolist()
    assert relerr(out["comps"], g["comps"]) < TOL
    assert relerr(out["weigs"], g["weigs"]) < TOL
    assert relerr(out["measures"], g["measures"]) < TOL
    if support == "local":
        assert relerr(geo(int(g["geo_idx"][0])), g["geo_phi_first"]) < 1e-10


@pytest.mark.parametrize("name", POS_CASES)
def test_post_process_and_storage(name):
    g = load_golden(name)
    std = bool(g["param_q_standarize"])
    post = orc.post_process_components(
        g["comps"], float(g["pre_scale_factor"]) if std else None, g["mean"] if std else None,
        bool(g["param_q_orthogonal"]), g["invMassL"] if bool(g["param_q_massWeight"]) else None)
    assert relerr(post, g["comps_post"]) < 1e-11
    assert relerr(orc.bases_sing_vals(g["comps_post"]), g["bases_sing_vals"]) < 1e-11
    raw = orc.components_bin_bytes(g["comps_post"])
    assert raw == g["bin_bytes"].tobytes()
    assert hashlib.sha256(raw).hexdigest() == str(g["bin_sha256"])
    K, F = g["comps"].shape[0], g["verts"].shape[0]
    assert orc.components_bin_name("q_pos_", F, K) == str(g["bin_name"])
    assert orc.components_npy_name("q_pos_", F, K) == str(g["npy_name"])
    if bool(g["param_q_orthogonal"]):
        M = orc.utmu(g["comps_post"], g["mass"])
        assert np.allclose(M, np.eye(K)[None])


def test_splocs_trace_and_components():
    g = load_golden("splocs_small")
    pre = _prep(g)
    geo = orc.Geodesics(g["verts"][0], g["tris"])
    K = int(g["param_vertPos_numComponents"])
    dmin, dmax = float(g["param_vertPos_smooth_min_dist"]), float(g["param_vertPos_smooth_max_dist"])
    d = orc.extract_k_components(pre["snapTensor"], K, "local", geo, dmin, dmax)
    s = orc.splocs_glob_optimization(pre["snapTensor"], d["comps"], d["weigs"], d["R"], geo, dmin, dmax,
                                     int(g["param_splocs_max_itrs"]), int(g["param_splocs_admm_num_itrs"]),
                                     float(g["param_splocs_lambda"]), float(g["param_splocs_rho"]))
    # the reference prints with %f (6 decimals): compare parsed floats, SURVEY.md 8(d)
    assert np.allclose(s["trace"], g["splocs_trace"], rtol=1e-8, atol=1e-6)
    assert s["idx"].tolist() == g["splocs_centres"].tolist()
    assert relerr(s["C"], g["splocs_C_final"]) < 1e-10
    assert relerr(s["Lambda"], g["splocs_Lambda_final"]) < 1e-10
    # the reference leaves comps/weigs untouched by SPLOCS
    assert relerr(d["comps"], g["comps"]) < TOL


def test_csv_text():
    g = load_golden("pca_global_small")
    lines = str(g["csv_text"]).splitlines()
    assert lines[0] == "component,singVal,norm_R"
    rows = np.array([[float(x) for x in ln.split(",")] for ln in lines[1:] if ln])
    assert relerr(rows, g["measures"]) < 1e-12


@pytest.mark.parametrize("name", ["pod_deim_small", "pod_deim_small_qr"])
def test_pod_deim(name):
    g = load_golden(name)
    pre = orc.prepare_nonlinear_snapshots(g["frames"], "first", True)
    assert relerr(pre["snapTensor"], g["snapTensor"]) < TOL
    K = int(g["K"])
    pod = orc.pod_vectorized(pre["snapTensor"], K)
    assert relerr(pod["S"], g["S"]) < 1e-11
    assert relerr(pod["comps"], g["comps"]) < 1e-9
    post, snap_post = orc.post_process_constraint_components(
        g["comps"], pre["snapTensor"], pre["pre_scale_factor"], pre["mean"], bool(g["orthogonal"]))
    assert relerr(post, g["comps_post"]) < 1e-11
    assert relerr(snap_post, g["snapTensor_post"]) < 1e-12
    dm = orc.deim(g["comps_post"], int(g["p"]))
    assert dm["Pt"].tolist() == g["Pt"].tolist()
    assert dm["alpha"].tolist() == g["alpha"].tolist()
    assert dm["alpha_ranges"].tolist() == g["alpha_ranges"].tolist()


@pytest.mark.parametrize("p", [1, 3])
def test_pca_blocks(p):
    """'pca_blocks' constraint bases (constraintsComponents.py:324-412) against the unmodified reference."""
    g = load_golden("pca_blocks_p%d" % p)
    pre = orc.prepare_nonlinear_snapshots(g["frames"], "first", True)
    assert relerr(pre["snapTensor"], g["snapTensor"]) < TOL
    r = orc.pca_blocks(pre["snapTensor"], int(g["K"]), p)
    assert r["points"].tolist() == g["points"].tolist()
    assert r["blocks"].tolist() == g["blocks"].tolist()
    sgn = np.sign(np.sum(r["weigs"] * g["weigs"], axis=0))          # LAPACK's SVD sign per component
    assert relerr(r["weigs"] * sgn[None], g["weigs"]) < 1e-10
    assert relerr(r["comps"] * sgn[:, None, None], g["comps"]) < 1e-10
    assert relerr(r["measures"], g["measures"]) < 1e-10
