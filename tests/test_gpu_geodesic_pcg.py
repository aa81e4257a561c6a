"""GPU (-m gpu): the sparse mode of the device geodesics (two-level PCG: Jacobi + aggregate coarse space,
csrc/asb_geodesic.hip) against the reference's algorithm with SuperLU (oracle.Geodesics, utils/support.py:139-208) -- on
the bunny (where Jacobi alone needed thousands of iterations) and on a mesh ABOVE the dense mode's 46 000 vertices, where
it is what `support='local'` uses by default."""
import types

import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import asb_oracle as orc

pytestmark = pytest.mark.gpu


def test_two_level_pcg_on_the_bunny_vs_superlu():
    from animsnapbases_amd import GeodesicDistanceComputation, HipEngine
    g = load_golden("c2_bunny_pca_global")
    V, T = g["rest"], g["tris"].astype(np.int64)
    eng = HipEngine(0)
    geo = GeodesicDistanceComputation(V, T, engine=eng, backend="pcg")
    ref = orc.Geodesics(V, T)
    src = [0, 123, 7777, 14289]
    phi = geo.solve_many(src)
    its = geo.last_iterations[0]
    assert its[0] < 20000 and its[1] < 400, its                 # (heat, Poisson) iterations; Jacobi alone: thousands
    for q, s in enumerate(src):
        assert relerr(phi[q], ref(s)) < 1e-8
    assert geo.n_aggregates > 500
    eng.close()


def _torus(nu, nv, R=0.4, r=0.15):
    """A quasi-uniform triangle mesh with nu * nv vertices (a regular grid bent into a torus)."""
    th, ph = 2 * np.pi * np.arange(nu) / nu, 2 * np.pi * np.arange(nv) / nv
    T, P = np.meshgrid(th, ph, indexing="ij")
    V = np.stack([(R + r * np.cos(P)) * np.cos(T), (R + r * np.cos(P)) * np.sin(T), r * np.sin(P)], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a, b = (i * nv + j).ravel(), (((i + 1) % nu) * nv + j).ravel()
    c, d = (i * nv + (j + 1) % nv).ravel(), (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    return V, np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.int64)


@pytest.mark.parametrize("mode", ["auto", "device"])
def test_local_support_above_the_dense_limit_uses_the_sparse_solver(mode, monkeypatch):
    """47 500 vertices: posSnapshots' automatic choice is the SLAB mode (round 4: direct block-tridiagonal factorisation; dense
    inverses stop at 46 000), ASB_GEODESIC=device still gives round 2's two-level PCG; either way the local-support deflation
    must give the oracle's sequence and basis (oracle: SuperLU on the host)."""
    monkeypatch.setenv("ASB_GEODESIC", mode)
    from animsnapbases_amd import posComponents, posSnapshots
    rest, tris = _torus(475, 100)
    assert rest.shape[0] == 47500
    verts = orc.synth_snapshots(rest, 12, rank=4, seed=5, kind="bumps")
    K = 3
    param = types.SimpleNamespace(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                                  q_support="local", vertPos_numComponents=K, store_vertPos_PCA_sing_val=False,
                                  vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.3, vertPos_rest_shape="first",
                                  name="t", vertPos_output_directory=".")
    snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False)
    geo = snaps.compute_geodesic_distance
    assert geo._engine is not None
    comp = posComponents(param, snaps)
    comp.compute_components_store_singvalues()
    if mode == "auto":
        assert snaps._engine.geodesic_dense and geo.n_slabs >= 8      # the whole local step stays on the device
        print("slabs:", geo.n_slabs, "largest", geo.largest_slab)
    else:
        assert not snaps._engine.geodesic_dense
        print("sweeps / PCG iterations of the last field:", geo.last_iterations)
    pre = orc.prepare_snapshots(verts, "first", True)
    ref_geo = orc.Geodesics(verts[0], tris)
    d = orc.extract_k_components(pre["snapTensor"], K, "local", ref_geo, 0.1, 0.3)
    assert comp.selected_vertices.tolist() == d["idx"].tolist()
    assert relerr(geo(int(d["idx"][0])), ref_geo(int(d["idx"][0]))) < 1e-8
    assert relerr(comp.comps, d["comps"]) < 1e-7 and relerr(comp.weigs, d["weigs"]) < 1e-9


def test_slab_mode_on_the_bunny_vs_superlu():
    from animsnapbases_amd import GeodesicDistanceComputation, HipEngine
    g = load_golden("c2_bunny_pca_global")
    V, T = g["rest"], g["tris"].astype(np.int64)
    eng = HipEngine(0)
    geo = GeodesicDistanceComputation(V, T, engine=eng, backend="slab")
    ref = orc.Geodesics(V, T)
    src = [0, 123, 7777, 14289]
    phi = geo.solve_many(src)
    print("bunny: %d slabs, largest %d" % (geo.n_slabs, geo.largest_slab))
    for q, s in enumerate(src):
        assert relerr(phi[q], ref(s)) < 1e-9, (s, relerr(phi[q], ref(s)))
    eng.close()


def test_slab_mode_solves_the_badly_graded_mesh():
    """The lat-long sphere with 310 slivers around each pole (47 122 vertices, element sizes over orders of magnitude) that the
    Jacobi-sweep heat step refuses (next test): the slab mode is a direct factorisation and has no convergence question --
    distances from a pole, from the equator and from a sliver vertex against SuperLU on the host (utils/support.py:173-208)."""
    from animsnapbases_amd import GeodesicDistanceComputation, HipEngine
    V, T = orc.synth_mesh(152, 310, seed=5)
    eng = HipEngine(0)
    geo = GeodesicDistanceComputation(V, T, engine=eng, backend="slab")
    ref = orc.Geodesics(V, T)
    src = [0, 100, V.shape[0] // 2, V.shape[0] - 1]
    phi = geo.solve_many(src)
    print("sliver sphere: %d vertices, %d slabs, largest %d" % (V.shape[0], geo.n_slabs, geo.largest_slab))
    for q, s in enumerate(src):
        err = relerr(phi[q], ref(s))
        print("source %d: relative error %.2e" % (s, err))
        assert err < 1e-8
    eng.close()


def test_sparse_mode_fails_loudly_on_a_badly_graded_mesh():
    """A lat-long sphere with 310 slivers around each pole: element sizes differ by orders of magnitude and the heat
    step's sweeps cannot converge -- the solve must say so (and name the host backend), not return distances."""
    from animsnapbases_amd import GeodesicDistanceComputation, HipEngine
    V, T = orc.synth_mesh(152, 310, seed=5)
    eng = HipEngine(0)
    geo = GeodesicDistanceComputation(V, T, engine=eng, backend="pcg")
    with pytest.raises(RuntimeError, match="ASB_GEODESIC=host"):
        geo(100)
    eng.close()
