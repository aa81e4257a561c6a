"""CPU: host-side logic of the product that does not need the GPU -- file formats,
geodesics, partitioning -- against the golden vectors / the oracle."""
import hashlib
import os

import numpy as np
import pytest

import types

from conftest import GOLDEN, load_golden, relerr
from animsnapbases_amd import GeodesicDistanceComputation, partition
from animsnapbases_amd import utils as U
from animsnapbases_amd.posComponents import posComponents
from oracle import asb_oracle as orc


@pytest.mark.parametrize("name", ["pca_global_small", "pca_global_avg_mass_orth", "pca_local_small"])
def test_bin_bytes_match_reference(name, tmp_path):
    g = load_golden(name)
    comps = g["comps_post"]
    K, N, _ = comps.shape
    F = g["verts"].shape[0]
    prefix = str(tmp_path / "q_pos_")
    U.store_components(prefix, F, K, N, 3, comps, ".bin", "K")
    U.store_components(prefix, F, K, N, 3, comps, ".npy", "K")
    raw = open(str(tmp_path / str(g["bin_name"])), "rb").read()
    assert hashlib.sha256(raw).hexdigest() == str(g["bin_sha256"])
    assert np.array_equal(np.load(str(tmp_path / str(g["npy_name"]))), comps)
    assert np.array_equal(U.read_components_bin(str(tmp_path / str(g["bin_name"]))), comps)


def test_geodesics_match_reference_and_oracle():
    g = load_golden("pca_local_small")
    geo = GeodesicDistanceComputation(g["verts"][0], g["tris"])
    phis = g["geo_phi_deflation"]
    for i, idx in enumerate(g["geo_idx"][:phis.shape[0]]):
        assert relerr(geo(int(idx)), phis[i]) < 1e-9
    many = geo.solve_many(g["geo_idx"][:phis.shape[0]])
    assert relerr(many, phis) < 1e-9
    o = orc.Geodesics(g["verts"][0], g["tris"])
    assert relerr(geo(7), o(7)) < 1e-9


def test_statics_match_oracle():
    rng = np.random.default_rng(3)
    x = rng.normal(size=50)
    assert np.array_equal(posComponents.project_weight(x), orc.project_weight(x))
    assert np.array_equal(posComponents.project_weight(-np.abs(x)), np.zeros(50))
    Lam = rng.uniform(0, 2, size=(4, 9))
    c = rng.normal(size=(4, 9, 3))
    c[1, 2] = 0.0          # zero-length group: shrink * 0 stays 0, no NaN
    assert np.array_equal(posComponents.prox_l1l2(Lam, c, 0.1), orc.prox_l1l2(Lam, c, 0.1))


def test_partition():
    assert partition(10, 1) == [(0, 10)]
    assert partition(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    sh = partition(100000, 8)
    assert sum(n for _, n in sh) == 100000 and all(sh[i][0] + sh[i][1] == sh[i + 1][0] for i in range(7))


def test_mass_bin_roundtrip(tmp_path):
    m = np.random.default_rng(0).uniform(0.1, 1, 17)
    p = str(tmp_path / "m.bin")
    U.write_mass_bin(p, m)
    assert np.array_equal(U.read_mass_bin(p, 17), m)
    assert np.array_equal(orc.read_mass_bin(p, 17), m)


def test_voronoi_masses_sum_to_area():
    V, T = orc.synth_mesh(8, 12, seed=0)
    m = U.voronoi_vertex_masses(V, T)
    area = 0.5 * np.linalg.norm(np.cross(V[T[:, 1]] - V[T[:, 0]], V[T[:, 2]] - V[T[:, 0]]), axis=1).sum()
    assert abs(m.sum() - area) < 1e-12 * area and (m > 0).all()


def test_rank_diagnostic_from_gram_matrices(capsys):
    """The printed rank check (utils/utils.py:60-74) driven by per-dimension Gram matrices must say what the
    row-by-row version says: full rank from the Gram rule, anything doubtful decided on the host slice."""
    rng = np.random.default_rng(3)
    n, K = 400, 7
    A = rng.normal(size=(n, K, 3))
    A[:, 5, 1] = A[:, 2, 1] * 2.0                     # dimension 1: rank K - 1
    G = np.stack([A[:, :, j].T @ A[:, :, j] for j in range(3)])
    asked = []

    def host_slice(j):
        asked.append(j)
        return A[:, :, j]

    U.test_linear_dependency_grams(G, K, host_slice)
    by_gram = capsys.readouterr().out.splitlines()
    U.test_linear_dependency(A, 3, K)
    by_rows = capsys.readouterr().out.splitlines()
    assert by_gram == by_rows
    assert by_gram[0] == ".. linear independent." and "rank: %d" % (K - 1) in by_gram[1]
    assert asked == [1]                               # only the doubtful dimension went to the host


def test_constraint_path_host_helpers(tmp_path):
    """Host pieces of SURVEY 8 f-4 / a15 that need no GPU: the element lists around a vertex (utils/support.py:210-258, as the
    reference's Python loops give them), the sparse-operator readers (utils/utils.py:289-323), the mass FILE branch of
    nonlinearSnapshots.load_factorize_masses (nonlinear_snapshots.py:180-191, 244-262) and the per-frame .bin reader (:126-143)."""
    import struct
    import types
    from scipy import sparse
    from animsnapbases_amd import constraints as C
    rng = np.random.default_rng(3)
    tris = rng.integers(0, 12, size=(30, 3))
    for v in (0, 5, 11):
        want = [i for i, t in enumerate(tris) if any(q in {v} for q in t)]                  # the reference's loop
        assert C.elements_of_vertex(v, tris) == want
        star = set()
        for f in tris:
            if [v] in f:
                star.update(f)
        assert sorted(C.vertex_star([v], tris)) == sorted(int(q) for q in star)
    M = sparse.random(9, 14, density=0.3, random_state=4, format="csr")
    np.savez(str(tmp_path / "st.npz"), St=np.array(M, dtype=object))
    got = C.read_sparse_matrix(str(tmp_path / "st.npz"), ".npz", key="St")
    assert (got != M).nnz == 0
    coo = M.tocoo()
    with open(str(tmp_path / "st.bin"), "wb") as fh:
        fh.write(struct.pack("<iii", 9, 14, coo.nnz))
        for r, c, v in zip(coo.row, coo.col, coo.data):
            fh.write(struct.pack("<iid", int(r), int(c), float(v)))
    assert (C.read_sparse_matrix(str(tmp_path / "st.bin"), ".bin") != M).nnz == 0
    # masses
    mass = rng.uniform(0.5, 2.0, size=20)
    with open(str(tmp_path / "m.bin"), "wb") as fh:
        fh.write(struct.pack("<ii", 20, 1))
        fh.write(mass.astype("<f8").tobytes())
    param = types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=3, constProj_p_size=2,
                                  constProj_masses_file=str(tmp_path / "m.bin"), constProj_frame_increment=2,
                                  constProj_train_test_jump=1, constProj_input_snapshots_pattern=str(tmp_path / "snap_"))
    ns = C.nonlinearSnapshots(param)
    ns.config()
    ns.load_factorize_masses()
    assert np.array_equal(ns.mass, mass) and np.allclose(ns.massL ** 2, mass) and np.allclose(ns.invMassL * ns.massL, 1.0)
    # per-frame .bin files: <i rows><i cols>, column by column
    frames = rng.normal(size=(6, 20, 3))
    for i in range(6):
        with open(str(tmp_path / ("snap_%d.bin" % i)), "wb") as fh:
            fh.write(struct.pack("<ii", 20, 3))
            fh.write(frames[i].T.astype("<f8").tobytes())
    X = ns.read(".bin")
    assert np.array_equal(X, frames[0:6:2]) and np.array_equal(ns.test_snapTensor, frames[1:6:2])
    assert ns.num_constained_elements == 10


# ---------------------------------------------------------------------------------------------------------------------
# masses derived from a mesh (posSnapshots.py:130-139, nonlinear_snapshots.py:192-240)
# ---------------------------------------------------------------------------------------------------------------------
def test_mesh_masses_against_the_reference_functions(tmp_path):
    """The reference's own arithmetic (utils/support.py:12-76, utils/utils.py:325-389), recorded by oracle/gen_golden.py
    meshmass: the .mesh reader, the normalised tetrahedral lumping, the per-element sums."""
    from animsnapbases_amd import utils as u
    g = np.load(os.path.join(GOLDEN, "mesh_masses.npz"))
    path = tmp_path / "cubes.mesh"
    path.write_bytes(g["mesh_text"].tobytes())
    V, T, tris = u.read_mesh_file(str(path))
    assert np.array_equal(V, g["read_V"]) and np.array_equal(T, g["read_T"]) and np.array_equal(tris, g["read_tris"])
    vm = u.lumped_tet_vertex_masses(V, T)
    assert np.allclose(vm, g["lumped_vertex_mass"], rtol=1e-13, atol=0) and abs(vm.sum() - 1.0) < 1e-14
    assert np.allclose(u.element_masses(vm, T, 3), g["tet_masses"], rtol=1e-14, atol=0)
    assert np.allclose(u.element_masses(vm, g["edges"], 1), g["edge_masses"], rtol=1e-14, atol=0)
    assert np.allclose(u.element_masses(vm, tris, 2), g["tri_masses"], rtol=1e-14, atol=0)


def test_vertex_masses_closed_forms():
    """What the reference takes from libigl (absent here: "parity unpinned"), against closed forms of the definitions."""
    from animsnapbases_amd import utils as u
    # a unit cube cut into six tetrahedra around its main diagonal: the masses sum to the volume, corner shares are known
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]], dtype=float)
    T = np.array([[0, 1, 3, 7], [0, 3, 2, 7], [0, 2, 6, 7], [0, 6, 4, 7], [0, 4, 5, 7], [0, 5, 1, 7]])
    assert np.allclose(u.tet_volumes(V, T), 1.0 / 6.0)
    m = u.tet_barycentric_vertex_masses(V, T)
    assert abs(m.sum() - 1.0) < 1e-15
    assert np.allclose(m[[0, 7]], 6 / 24.0) and np.allclose(m[1:7], 2 / 24.0)      # the diagonal's ends touch all six, the rest two
    A = np.array([[2.0, 0.3, 0.1], [0.0, 1.5, 0.2], [0.1, 0.0, 0.7]])                # an affine map scales every volume by |det|
    assert np.allclose(u.tet_barycentric_vertex_masses(V @ A.T, T), m * abs(np.linalg.det(A)))
    # a regular grid of right triangles (non-obtuse): the mixed-Voronoi cells tile the plane
    n = 6
    gx, gy = np.meshgrid(np.arange(n, dtype=float), np.arange(n, dtype=float), indexing="ij")
    P = np.stack([gx.ravel(), gy.ravel(), np.zeros(n * n)], 1)
    vid = lambda i, j: i * n + j
    tris = np.array([t for i in range(n - 1) for j in range(n - 1)
                     for t in ([vid(i, j), vid(i + 1, j), vid(i + 1, j + 1)], [vid(i, j), vid(i + 1, j + 1), vid(i, j + 1)])])
    mv = u.voronoi_vertex_masses(P, tris)
    assert abs(mv.sum() - (n - 1) ** 2) < 1e-12                                     # total area
    assert np.allclose(mv.reshape(n, n)[1:-1, 1:-1], 1.0)                            # an interior vertex owns one unit cell
    # an obtuse triangle: half of the area to the obtuse corner, a quarter to the others
    Pt = np.array([[0, 0, 0], [4, 0, 0], [2, 0.5, 0]], dtype=float)
    mo = u.voronoi_vertex_masses(Pt, np.array([[0, 1, 2]]))
    assert np.allclose(mo, [0.25, 0.25, 0.5])
    # edges of a tetrahedron, in libigl's order (by the larger end point, then the smaller)
    assert u.mesh_edges(np.array([[0, 1, 2, 3]])).tolist() == [[0, 1], [0, 2], [1, 2], [0, 3], [1, 3], [2, 3]]


def test_element_masses_from_the_mesh_feed_the_constraint_snapshots(tmp_path):
    """nonlinear_snapshots.py:192-240 end to end on the host: no mass file -> masses from the mesh, one per constraint row."""
    from animsnapbases_amd import utils as u
    from animsnapbases_amd.constraints import nonlinearSnapshots
    g = np.load(os.path.join(GOLDEN, "mesh_masses.npz"))
    path = tmp_path / "cubes.mesh"
    path.write_bytes(g["mesh_text"].tobytes())
    ns = nonlinearSnapshots.__new__(nonlinearSnapshots)
    ns.param = types.SimpleNamespace(tet_mesh_file=str(path), tri_mesh_file=None, volumetric_mesh=True,
                                     constProj_snapshots_type="edge_spring")
    ns._preset_mass, ns.mass_file = None, ""
    T = g["read_T"]
    # p = 3: tetrahedral strain, three rows per tetrahedron
    ns.constraintsSize, ns.num_constained_elements = 3, T.shape[0]
    ns.load_factorize_masses()
    vm = u.tet_barycentric_vertex_masses(g["read_V"], T)
    assert ns.mass.shape == (3 * T.shape[0],) and np.allclose(ns.mass[::3], vm[T].sum(1)) and np.allclose(ns.mass[1::3], ns.mass[::3])
    assert np.allclose(ns.massL ** 2, ns.mass) and np.allclose(ns.massL * ns.invMassL, 1.0)
    # p = 1 on the volumetric mesh: edge springs, the reference's normalised lumping
    e = u.mesh_edges(T)
    ns.constraintsSize, ns.num_constained_elements = 1, e.shape[0]
    ns.load_factorize_masses()
    assert np.allclose(ns.mass, g["lumped_vertex_mass"][e].sum(1))


# ---------------------------------------------------------------------------------------------------------------------
# the recorder's binary files, written here with struct exactly as the reference READS them element by element
# (nonlinear_snapshots.py:126-160, :180-191; utils/utils.py:289-311)
# ---------------------------------------------------------------------------------------------------------------------
def test_bin_readers_round_trip(tmp_path):
    import struct
    from animsnapbases_amd.constraints import nonlinearSnapshots, read_sparse_matrix
    rng = np.random.default_rng(3)
    F, ep, inc, jump = 5, 7, 2, 1
    frames = rng.normal(size=(F * inc, ep, 3))
    pattern = str(tmp_path / "snap_")
    for i in range(F * inc):                                   # <i rows><i cols>, then column by column
        with open(pattern + str(i) + ".bin", "wb") as fh:
            fh.write(struct.pack("<i", ep) + struct.pack("<i", 3))
            for c in range(3):
                for r in range(ep):
                    fh.write(struct.pack("<d", frames[i, r, c]))
    mass = rng.uniform(0.5, 2.0, size=ep)
    mass_file = str(tmp_path / "mass.bin")
    with open(mass_file, "wb") as fh:                          # <i n><i m>, then n doubles
        fh.write(struct.pack("<i", ep) + struct.pack("<i", 1))
        for v in mass:
            fh.write(struct.pack("<d", v))
    ns = nonlinearSnapshots.__new__(nonlinearSnapshots)
    ns.param = types.SimpleNamespace(constProj_frame_increment=inc, constProj_train_test_jump=jump)
    ns._frames = ns._test_frames = None
    ns.snapshots_file, ns.frs, ns.constraintsSize = pattern, F, 1
    X = ns.read(".bin")
    assert X.dtype == np.float64 and np.array_equal(X, frames[0:F * inc:inc])
    assert np.array_equal(ns.test_snapTensor, frames[jump:F * inc:inc])
    assert ns.num_constained_elements == ep
    ns._preset_mass, ns.mass_file = None, mass_file
    ns.load_factorize_masses()
    assert np.array_equal(ns.mass, mass) and np.allclose(ns.massL, np.sqrt(mass)) and np.allclose(ns.invMassL * ns.massL, 1.0)
    # sparse operator: <i rows><i cols><i nnz>, then nnz triplets <i row><i col><d value>; repeated entries add up (csr_matrix)
    rows, cols = 6, ep
    trip = [(int(rng.integers(rows)), int(rng.integers(cols)), float(rng.normal())) for _ in range(15)] + [(2, 3, 1.5), (2, 3, -0.25)]
    st_file = str(tmp_path / "St.bin")
    with open(st_file, "wb") as fh:
        fh.write(struct.pack("<i", rows) + struct.pack("<i", cols) + struct.pack("<i", len(trip)))
        for r, c, v in trip:
            fh.write(struct.pack("<i", r) + struct.pack("<i", c) + struct.pack("<d", v))
    dense = np.zeros((rows, cols))
    for r, c, v in trip:
        dense[r, c] += v
    St = read_sparse_matrix(st_file, ".bin")
    assert St.shape == (rows, cols) and np.allclose(St.toarray(), dense, rtol=0, atol=1e-15)
    with pytest.raises(ValueError):
        read_sparse_matrix(st_file, ".txt")


def test_constraints_diagnostics_against_the_reference(capsys):
    """constraintsComponents' diagnostics (constraintsComponents.py:452-487, 524-570: host NumPy on the downloaded basis) against
    what the unmodified reference returned / printed on the same basis (oracle/gen_golden.py diagnostics)."""
    from animsnapbases_amd.constraints import constraintsComponents
    g = np.load(os.path.join(GOLDEN, "constraints_diagnostics.npz"))
    cc = constraintsComponents.__new__(constraintsComponents)
    cc._comps, cc._comps_on_device = g["comps"].copy(), False
    cc.nonlinearSnapshots = types.SimpleNamespace(mass=g["mass"], constraintsSize=1, dim=3, snapTensor=g["snapTensor"])
    assert np.allclose(cc.matrix_properties_test(g["Pt"]), g["mat_e"], rtol=1e-9, atol=0)
    assert np.allclose(cc.test_basesSingVals(), g["bases_sing_vals"], rtol=1e-12, atol=1e-15)
    cc.is_utmu_orthogonal()
    assert capsys.readouterr().out == str(g["utmu_stdout"])
    assert abs(constraintsComponents.frobenius_error(g["snapTensor"], g["rec"]) - float(g["frobenius"])) < 1e-12 * float(g["frobenius"])
    assert np.allclose(constraintsComponents.relative_error_per_component(g["snapTensor"], g["rec"]), g["relative"], rtol=1e-13)
    assert abs(constraintsComponents.max_pointwise_error(g["snapTensor"], g["rec"]) - float(g["max_pointwise"])) < 1e-14
