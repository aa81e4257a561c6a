"""Host-side helpers on the hot path: file formats, the timing log, printed checks.

Mirrors (same names, argument meaning, file bytes and printed text) the few functions of
the reference's ``utils/utils.py`` that ``posSnapshots`` / ``posComponents`` call:
``store_components`` :14-38, ``testSparsity`` :41-57, ``test_linear_dependency`` :60-74,
``log_time`` :209-237.  Nothing here touches the GPU.
"""
import functools
import os
import struct
import time

import numpy as np
from numpy.linalg import matrix_rank


# ------------------------------------------------------------------ timing log
_has_written = False


class no_gc(object):
    """The host loops that feed the GPU launch by launch (SPLOCS' outer iterations, the per-component residual loop) leave the
    device idle for as long as the interpreter pauses: a generation-2 garbage collection in the middle of one costs 2 - 3 ms
    (measured on config 3: two such holes in 20 outer iterations).  Collection is deferred to the end of the loop."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def log_time(filePath):
    """Decorator: prints and logs ``Function '<name>' executed in X.XXXX seconds.`` to
    ``<filePath>function_timings.txt`` (first call of the process truncates, later calls
    append) -- the format of utils/utils.py:209-237, which main.py:81 later moves to
    ``<out>/time_logs.txt``."""

    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            global _has_written
            mode = "a" if _has_written else "w"
            _has_written = True
            t0 = time.time()
            result = func(*args, **kwargs)
            dt = time.time() - t0
            line = "Function '%s' executed in %.4f seconds." % (func.__name__, dt)
            print(line)
            try:
                with open(filePath + "function_timings.txt", mode) as fh:
                    fh.write(line + "\n")
            except OSError:
                pass        # a read-only cwd must not break the computation
            return result

        return wrapper

    return decorator


# ------------------------------------------------------------------ component files
def components_bin_bytes(basesTensor):
    """The exact byte image the reference writes with a Python triple loop
    (utils/utils.py:26-35): ``<i N><i 3K>`` then x-block, y-block, z-block, each K columns
    of N little-endian doubles (column-major N x 3K)."""
    K, N, dim = basesTensor.shape
    body = np.ascontiguousarray(np.transpose(basesTensor, (2, 0, 1)), dtype="<f8")
    return struct.pack("<i", N) + struct.pack("<i", dim * K) + body.tobytes()


def store_components(fileName, F, K, N, dim, basesTensor, extension='.bin', colName='K'):
    """utils/utils.py:14-38 (same names / files; the .bin body is written in one call)."""
    assert basesTensor.shape == (K, N, dim)
    if extension == '.bin':
        with open(fileName + 'F' + str(F) + colName + str(K) + extension, 'wb') as doc0:
            doc0.write(components_bin_bytes(basesTensor))
    if extension == '.npy':
        np.save(fileName + str(F) + 'K' + str(K), basesTensor)


def store_interpol_points_vector(fileName, F, K, points, extension='.bin', colName='K'):
    """utils/utils.py:77-98: ``<i n><i 1>`` + n doubles, or ``.npy``."""
    points = np.asarray(points)
    assert K <= points.shape[0]
    print("Storing ", points.shape[0], "interpolation points")
    if extension == '.bin':
        with open(fileName + 'F' + str(F) + colName + str(K) + "_points" + str(points.shape[0]) + extension, 'wb') as doc0:
            doc0.write(struct.pack("<ii", points.shape[0], 1))
            doc0.write(np.asarray(points, dtype="<f8").tobytes())
    if extension == '.npy':
        np.save(fileName + str(F) + 'K' + str(K) + "_points" + str(points.shape[0]), points)


def read_components_bin(path):
    """Inverse of ``store_components`` ('.bin'): returns (K, N, 3)."""
    with open(path, "rb") as fh:
        N, cols = struct.unpack("<ii", fh.read(8))
        body = np.frombuffer(fh.read(), dtype="<f8")
    K = cols // 3
    return np.ascontiguousarray(body.reshape(3, K, N).transpose(1, 2, 0))


# ------------------------------------------------------------------ K x K Gram matrices of the basis (rank diagnostic)
def summed_grams(eng, comm, K):
    """The three per-dimension K x K Gram matrices of the device-resident basis, summed over the ranks, on the host."""
    if comm.multi:
        Gbuf = comm.new_buffer(3 * K * K, eng.device_exchange)
        eng.orth_gram(Gbuf.data_ptr())
        comm.allreduce_tensor(Gbuf)
        G = Gbuf.cpu().numpy().reshape(3, K, K)
    else:
        eng.orth_gram(None)
        G = eng.orth_gram_get()
    return 0.5 * (G + G.transpose(0, 2, 1))


# ------------------------------------------------------------------ printed checks
def testSparsity(mat):
    """utils/utils.py:41-57."""
    assert mat.shape[2] == 3
    spar = [1 - (np.count_nonzero(mat[:, :, l]) / mat[:, :, l].size) for l in range(3)]
    if min(spar) > 0.5:
        print("sparse, min %" + str(100 * min(spar)) + " zero entries.")
    else:
        print("... not sparse.")


def _rank(A):
    """numpy.linalg.matrix_rank(A) (what the reference prints) at the cost of one Gram product when the answer is
    clear-cut: if the small Gram matrix has lambda_min / lambda_max > 1e-12, every singular value is above
    1e-6 sigma_max -- five orders over matrix_rank's tolerance max(M, N) eps sigma_max -- so the rank is full.
    Anything closer is decided by the SVD itself."""
    A = np.ascontiguousarray(A)          # strided slices would miss BLAS in the product below
    m = min(A.shape)
    if m == 0 or not np.all(np.isfinite(A)):
        return matrix_rank(A)
    G = A.T @ A if A.shape[0] >= A.shape[1] else A @ A.T
    lam = np.linalg.eigvalsh(G)
    if lam[-1] > 0 and lam[0] > 1e-12 * lam[-1]:
        return m
    return matrix_rank(A)


def test_linear_dependency(mat, test_dim_range, expected_rank):
    """utils/utils.py:60-74."""
    assert mat.shape[2] == 3
    for j in range(test_dim_range):
        rk = _rank(mat[:, :, j])
        if rk == expected_rank:
            print(".. linear independent.")
        else:
            print("... not linear independent, with rank: " + str(rk) + " != " + str(expected_rank) + ".")


def test_linear_dependency_grams(G, expected_rank, host_slice, lams=None):
    """The same printed check from the per-dimension Gram matrices G (3, K, K) of a device-resident basis (one MFMA
    product on the GPU instead of a host product over all rows): clear-cut cases by the rule of ``_rank``; anything
    closer is decided on the host, ``host_slice(j)`` supplying the (rows, K) slice.  ``lams``: the eigenvalues of the G[j], if the
    caller has them already (constraintsComponents.deim computes them beside the device loop)."""
    for j in range(G.shape[0]):
        lam = lams[j] if lams is not None else np.linalg.eigvalsh(G[j])
        if np.all(np.isfinite(lam)) and lam[-1] > 0 and lam[0] > 1e-12 * lam[-1]:
            rk = G.shape[1]
        else:
            rk = _rank(host_slice(j))
        if rk == expected_rank:
            print(".. linear independent.")
        else:
            print("... not linear independent, with rank: " + str(rk) + " != " + str(expected_rank) + ".")


# ------------------------------------------------------------------ masses
def read_mass_bin(fileName, N):
    """snapbases/posSnapshots.py:142-149: ``<i n><i m>`` + n doubles."""
    with open(fileName, "rb") as fh:
        ni, _mi = struct.unpack("<ii", fh.read(8))
        assert ni == N
        return np.frombuffer(fh.read(8 * N), dtype="<f8").astype(np.float64)


def write_mass_bin(fileName, mass):
    with open(fileName, "wb") as fh:
        fh.write(struct.pack("<ii", len(mass), 1))
        fh.write(np.asarray(mass, dtype="<f8").tobytes())


def voronoi_vertex_masses(verts, tris):
    """Mixed-Voronoi lumped vertex areas (what libigl's MASSMATRIX_TYPE_VORONOI computes,
    used by snapbases/posSnapshots.py:137 when no mass file exists): circumcentric cells
    for non-obtuse triangles; an obtuse triangle gives half its area to the obtuse corner
    and a quarter to the other two.  libigl is not in this image, so this branch is pinned
    only by its definition."""
    verts = np.asarray(verts, dtype=np.float64)
    n = verts.shape[0]
    p = [verts[tris[:, i]] for i in range(3)]
    l2 = [((p[(i + 1) % 3] - p[(i + 2) % 3]) ** 2).sum(axis=1) for i in range(3)]   # opposite edge^2
    area = 0.5 * np.sqrt((np.cross(p[1] - p[0], p[2] - p[0]) ** 2).sum(axis=1))
    cot = [(l2[(i + 1) % 3] + l2[(i + 2) % 3] - l2[i]) / (4 * area) for i in range(3)]
    obtuse = [c < 0 for c in cot]
    any_obtuse = obtuse[0] | obtuse[1] | obtuse[2]
    mass = np.zeros(n)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        vor = (l2[j] * cot[j] + l2[k] * cot[k]) / 8
        mixed = np.where(obtuse[i], area / 2, area / 4)
        mass += np.bincount(tris[:, i], np.where(any_obtuse, mixed, vor), minlength=n)
    return mass


def read_mesh_file(path):
    """Medit ``.mesh`` text file -> (vertices (n,3) float, tetrahedra (m,4) int, triangles (t,3) int), zero-based -- what
    utils/utils.py:325-389 of the reference returns: sections ``Vertices`` / ``Tetrahedra`` / ``Triangles``, each followed by a
    count line, every entry with a trailing reference number that is dropped; other sections are skipped."""
    want = {"Vertices": (3, float), "Tetrahedra": (4, int), "Triangles": (3, int)}
    got = {k: [] for k in want}
    with open(path, "r") as fh:
        tokens = iter(fh.read().split("\n"))
        for line in tokens:
            key = line.strip()
            name = next((k for k in want if key.startswith(k)), None)
            if name is None:
                continue
            count = int(next(tokens).strip())
            ncol = want[name][0]
            for _ in range(count):
                parts = next(tokens).split()
                if len(parts) >= ncol + 1 or (name != "Vertices" and len(parts) >= 4):
                    got[name].append(parts[:ncol] if name == "Vertices" else parts[:-1])
    V = np.array(got["Vertices"], dtype=float) if got["Vertices"] else np.array([], dtype=float)
    T = np.array(got["Tetrahedra"], dtype=int) - 1 if got["Tetrahedra"] else np.array([], dtype=int)
    F = np.array(got["Triangles"], dtype=int) - 1 if got["Triangles"] else np.array([], dtype=int)
    return V, T, F


def tet_volumes(V, T):
    a, b, c, d = (V[T[:, i]] for i in range(4))
    return np.abs(np.einsum("ij,ij->i", np.cross(b - a, c - a), d - a)) / 6.0


def tet_barycentric_vertex_masses(V, T):
    """A quarter of every tetrahedron's volume to each of its corners: the diagonal of libigl's mass matrix of a tetrahedral
    mesh (``igl.massmatrix(V, T)``: barycentric lumping is the only kind libigl builds for tetrahedra), which
    snapbases/posSnapshots.py:135 and nonlinear_snapshots.py:236 take.  libigl is not in this image: pinned by its definition
    and by closed forms (the masses sum to the mesh volume; tests/test_host_logic_cpu.py), "parity unpinned" in DESIGN.md."""
    V, T = np.asarray(V, dtype=np.float64), np.asarray(T, dtype=np.int64)
    return np.bincount(T.ravel(), np.repeat(tet_volumes(V, T) / 4.0, 4), minlength=V.shape[0])


def lumped_tet_vertex_masses(V, T, density=1.0):
    """utils/support.py:41-59 of the reference (``compute_lumped_mass_matrix``): barycentric lumping, then normalised to a total
    mass of one; returned as the vector of the diagonal."""
    m = density * tet_barycentric_vertex_masses(V, T)
    tot = m.sum()
    return m / tot if tot > 0 else m


def element_masses(vertex_masses, elements, aux_size):
    """Per auxiliary row of every element (edge, triangle, tetrahedron): the sum of its corners' masses, repeated ``aux_size``
    times -- utils/support.py:12-76 (``compute_edgeMasses`` / ``compute_triMasses`` / ``compute_tetMasses``) in one line."""
    vm = np.asarray(vertex_masses, dtype=np.float64)
    w = vm[np.asarray(elements, dtype=np.int64)].sum(axis=1)
    return np.repeat(w, int(aux_size))


def mesh_edges(simplices):
    """Unique undirected edges (i < j) of a triangle or tetrahedron list, ordered by (j, i) -- the order libigl's ``igl.edges``
    walks its adjacency matrix in (column by column, rows above the diagonal).  No libigl here: parity unpinned."""
    S = np.asarray(simplices, dtype=np.int64)
    k = S.shape[1]
    e = np.concatenate([S[:, [a, b]] for a in range(k) for b in range(a + 1, k)])
    e = np.unique(np.sort(e, axis=1), axis=0)
    return e[np.lexsort((e[:, 0], e[:, 1]))]


def read_triangle_mesh(path):
    """(vertices, triangles) of an ``.obj`` or ``.off`` file (what the reference reads with ``igl.read_triangle_mesh``)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".off":
        from .process import load_off
        v, f = load_off(path, no_colors=True)[:2]
        return np.asarray(v, dtype=float), np.asarray(f, dtype=np.int64)
    if ext == ".obj":
        vs, fs = [], []
        with open(path, "r") as fh:
            for line in fh:
                p = line.split()
                if not p:
                    continue
                if p[0] == "v":
                    vs.append([float(x) for x in p[1:4]])
                elif p[0] == "f":
                    idx = [int(q.split("/")[0]) - 1 for q in p[1:]]
                    for t in range(1, len(idx) - 1):          # fan triangulation of polygons
                        fs.append([idx[0], idx[t], idx[t + 1]])
        return np.array(vs, dtype=float), np.array(fs, dtype=np.int64)
    raise ValueError("read_triangle_mesh: unknown mesh file type " + path)


# ------------------------------------------------------------------ animation files
def read_animation(path):
    """``verts`` (F,N,3) and ``tris`` (M,3) from ``.h5`` (reference format,
    snapbases/posSnapshots.py:109-111; needs h5py) or ``.npz`` with the same keys."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        with np.load(path) as d:
            return d["verts"].astype(float), d["tris"]
    try:
        import h5py
    except ImportError as e:
        raise ImportError("reading %s needs h5py (not installed); store the animation as .npz with keys "
                          "'verts' (F,N,3) and 'tris' (M,3) instead" % path) from e
    with h5py.File(path, "r") as f:
        return f["verts"][()].astype(float), f["tris"][()]
