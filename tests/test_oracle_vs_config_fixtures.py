"""CPU: the NumPy oracle against the full-size config-2 / config-3 fixtures of the unmodified reference
(oracle/gen_golden_configs.py) -- pins the oracle at the sizes BASELINE.json names, not only on toy meshes.
Config 3 is checked on its first 6 components (the oracle's 64 x 1000-frame local deflation + 20 SPLOCS iterations
take the reference's ~15 minutes; the GPU test covers all of it)."""
import numpy as np
import pytest

from conftest import load_golden
from config_fixtures import check_deflation, regen_frames
from oracle import asb_oracle as orc


@pytest.mark.parametrize("name,K,signed", [("c2_bunny_pca_global", 32, False), ("c2_bunny_pca_local", 32, True),
                                           ("c3_armadillo_splocs", 6, True)])
def test_oracle_vs_full_size_fixture(name, K, signed):
    g = load_golden(name)
    verts = regen_frames(g)
    pre = orc.prepare_snapshots(verts, "first", True)
    geo = orc.Geodesics(verts[0], g["tris"].astype(np.int64)) if signed else None
    d = orc.extract_k_components(pre["snapTensor"], K, "local" if signed else "global", geo,
                                 float(g["param_vertPos_smooth_min_dist"]), float(g["param_vertPos_smooth_max_dist"]))
    check_deflation(g, pre["pre_scale_factor"], pre["mean"], d["idx"], d["comps"], d["weigs"], d["measures"],
                    signed=signed, tol=1e-10, mtol=1e-10)
