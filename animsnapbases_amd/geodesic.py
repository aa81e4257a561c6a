"""Heat-method geodesic distances -- the support maps of ``support='local'`` and SPLOCS.

Follows the formulas of the reference's ``GeodesicDistanceComputation`` (utils/support.py:139-208; cotan
Laplacian :81-136; m = 10, t = m*h^2), organised differently: the per-triangle gradient and the per-vertex
divergence are assembled ONCE as sparse matrices G (3M x N) and D (N x 3M); ``solve_many`` answers a batch of
sources at a time and remembers every field it has solved (SPLOCS asks for the same centres again once they settle).

Backends (chosen by posSnapshots, ``ASB_GEODESIC``):
* ``engine=..., backend="dense"`` (default): the two SPD systems the reference factorises with SuperLU (:170-171)
  are INVERTED once on the device (csrc/asb_dense.hip: blocked Gauss-Jordan on f64 MFMA); a query is a column gather
  for the heat step, two SpMMs for gradient / divergence and one dense product for the Poisson step.
* ``engine=None``: host SciPy SuperLU like the reference (symmetric minimum-degree ordering) -- opt-in only.
* ``engine=..., backend="pcg"``: sparse batched PCG on the device (csrc/asb_geodesic.hip), 64 sources at a time, with a
  two-level preconditioner: Jacobi + a piecewise-constant coarse space of ~50-vertex aggregates (built here, once) whose
  coarse operators are inverted densely on the device.  What meshes above the dense mode's 46 000 vertices use.
"""
import numpy as np
from scipy import sparse
from scipy.sparse.linalg import splu


def _vlen(v):
    return np.sqrt((v * v).sum(axis=-1))


def _unit(v):
    return v / _vlen(v)[..., None]


def cotan_laplacian(verts, tris):
    """L (negative semi-definite cotan Laplacian, rows sum to 0) and the lumped vertex
    areas (utils/support.py:81-136)."""
    n = verts.shape[0]
    I, J, Wv = [], [], []
    for a, b, c in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        pa, pb, pc = verts[tris[:, a]], verts[tris[:, b]], verts[tris[:, c]]
        u, v = pb - pa, pc - pa
        w = 0.5 * (u * v).sum(axis=1) / _vlen(np.cross(u, v))
        I += [tris[:, b], tris[:, c]]
        J += [tris[:, c], tris[:, b]]
        Wv += [w, w]
    Wm = sparse.csr_matrix((np.concatenate(Wv), (np.concatenate(I), np.concatenate(J))), shape=(n, n))
    L = (Wm - sparse.diags(np.asarray(Wm.sum(axis=1)).ravel())).tocsr()
    tri_area = 0.5 * _vlen(np.cross(verts[tris[:, 1]] - verts[tris[:, 0]], verts[tris[:, 2]] - verts[tris[:, 0]]))
    area = np.zeros(n)
    for i in range(3):
        area += np.bincount(tris[:, i], tri_area / 3, minlength=n)
    return L, area


def _greedy_aggregates(indptr, indices):
    """One pass of greedy aggregation on a graph in CSR form: a vertex whose whole 1-ring is still free becomes the root
    of a new aggregate (itself + the ring); what is left over joins a neighbouring aggregate.  Returns (labels, count)."""
    n = indptr.shape[0] - 1
    lab = np.full(n, -1, dtype=np.int64)
    na = 0
    for v in range(n):
        if lab[v] >= 0:
            continue
        nb = indices[indptr[v]:indptr[v + 1]]
        if nb.size and (lab[nb] >= 0).any():
            continue
        lab[v] = na
        lab[nb] = na
        na += 1
    for v in range(n):
        if lab[v] >= 0:
            continue
        nb = indices[indptr[v]:indptr[v + 1]]
        t = lab[nb]
        t = t[t >= 0]
        if t.size:
            lab[v] = t[0]
        else:
            lab[v] = na
            na += 1
    return lab, na


def mesh_aggregates(A, passes=None):
    """Aggregates of the graph of the sparse symmetric matrix A: the coarse space of the PCG mode's two-level
    preconditioner.  One greedy pass gives ~10 vertices per aggregate (bunny: 76 PCG iterations on the Poisson system
    against 595 with Jacobi alone), a second pass on the aggregate graph ~110 (212 iterations) but a 100 times smaller
    dense coarse problem: one pass while that stays below ~2000 aggregates, two above.
    Returns (agg (n,) int, number of aggregates)."""
    if passes is None:
        passes = 1 if A.shape[0] <= 20000 else 2
    G = sparse.csr_matrix((np.ones(A.nnz), A.indices.copy(), A.indptr.copy()), shape=A.shape)
    G.setdiag(0)
    G.eliminate_zeros()
    agg = np.arange(A.shape[0], dtype=np.int64)
    nc = A.shape[0]
    for _ in range(passes):
        lab, na = _greedy_aggregates(G.indptr, G.indices)
        agg = lab[agg]
        P = sparse.csr_matrix((np.ones(lab.shape[0]), (np.arange(lab.shape[0]), lab)), shape=(lab.shape[0], na))
        G = (P.T @ G @ P).tocsr()
        G.setdiag(0)
        G.eliminate_zeros()
        nc = na
        if nc < 64:
            break
    return agg, nc


def bfs_slabs(A, target=1536):
    """Breadth-first level sets of the graph of the sparse symmetric matrix A, grouped into slabs of at least ``target``
    vertices: a vertex's neighbours lie in its own or an adjacent level, so in the returned numbering A is BLOCK TRIDIAGONAL
    over the slabs (csrc/asb_geodesic.hip: the slab mode's direct factorisation).  The sweep starts at a pseudo-peripheral
    vertex (the last level of a sweep from vertex 0: long, narrow level structure); further components of a disconnected
    mesh follow as their own levels.  Returns (order (n,): permuted index -> vertex, ptr (nslab + 1,): slab boundaries)."""
    n = A.shape[0]
    indptr, indices = A.indptr, A.indices

    def sweep(start, seen):
        levels, cur = [], np.array([start], dtype=np.int64)
        seen[start] = True
        while cur.size:
            levels.append(cur)
            lo, hi = indptr[cur], indptr[cur + 1]
            nb = np.concatenate([indices[a:b] for a, b in zip(lo.tolist(), hi.tolist())]) if cur.size < 64 else \
                indices[np.concatenate([np.arange(a, b) for a, b in zip(lo.tolist(), hi.tolist())])]
            nb = np.unique(nb)
            nb = nb[~seen[nb]]
            seen[nb] = True
            cur = nb
        return levels
    seen = np.zeros(n, dtype=bool)
    first = sweep(0, seen)
    start = int(first[-1][0])
    seen[:] = False
    levels = sweep(start, seen)
    while not seen.all():                                   # other connected components
        levels += sweep(int(np.flatnonzero(~seen)[0]), seen)
    order, ptr, size = [], [0], 0
    for lv in levels:
        order.append(np.sort(lv))
        size += lv.size
        if size >= target:
            ptr.append(ptr[-1] + size)
            size = 0
    if size:
        if len(ptr) > 1 and size < target // 4:             # a small tail joins the last slab
            ptr[-1] += size
        else:
            ptr.append(ptr[-1] + size)
    return np.concatenate(order), np.asarray(ptr, dtype=np.int64)


def coarse_operators(A_heat, L, agg, nc):
    """Dense coarse matrices P^T (A - tL) P and P^T (-L) P + (gamma / nc) 1 1^T (the rank-one term fixes the constant null
    vector of the Laplacian, as in the dense mode), P = piecewise-constant prolongation of `agg`."""
    n = agg.shape[0]
    P = sparse.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
    Hc = np.asarray((P.T @ A_heat @ P).todense())
    Lc = np.asarray((P.T @ (-L) @ P).todense())
    gamma = np.trace(Lc) / nc
    Lc = Lc + gamma / nc
    return 0.5 * (Hc + Hc.T), 0.5 * (Lc + Lc.T)


class GeodesicDistanceComputation(object):
    """Callable: ``phi = geo(idx)`` -> (N,) geodesic distance from vertex ``idx``
    (shifted so that min(phi) == 0, utils/support.py:206)."""

    def __init__(self, verts, tris, m=10.0, engine=None, tol=1e-13, backend="pcg"):
        """engine: a HipEngine to run the solves on the GPU (``asb_geodesic_*``, 64 sources at a time); None = host
        SuperLU like the reference.  backend (with an engine): "dense" = the two SPD matrices are inverted once on the
        device (blocked Gauss-Jordan on f64 MFMA, N <= 46 000) and a query is a gather plus one dense product;
        "pcg" = batched Jacobi-PCG to relative residual ``tol`` (experimental)."""
        # LAZY (round 3): the reference factorises both systems eagerly inside posSnapshots (:96-99, 0.27 s of SuperLU on the
        # bunny) although only support='local' / SPLOCS ever query them; here construction only records its arguments and
        # ``prepare()`` -- operator assembly + the device set-up (two N x N inverses in the dense mode) -- runs on the first
        # query or the first access to one of the prepared attributes, so support='global' pays nothing.
        self._args = (np.asarray(verts, dtype=np.float64), np.asarray(tris, dtype=np.int64), m, engine, tol, backend)
        self.n = self._args[0].shape[0]
        self.last_iterations = None
        self._cache = {}                    # source vertex -> its distance field (solve_many)
        self.cache_bytes = 2 << 30
        self._ready = False

    _PREPARED = frozenset(("G", "D", "_A_heat", "_L", "_engine", "_heat", "_poisson", "n_aggregates", "_tol", "n_slabs",
                           "largest_slab"))

    def __getattr__(self, name):            # only reached for attributes that are not set (yet)
        if name in GeodesicDistanceComputation._PREPARED and not self.__dict__.get("_ready", True) \
                and not self.__dict__.get("_preparing", False):
            self.prepare()
            return getattr(self, name)
        raise AttributeError(name)

    def prepare(self):
        """Operator assembly and solver set-up (idempotent)."""
        if self._ready:
            return self
        if self.__dict__.get("_preparing"):
            raise RuntimeError("GeodesicDistanceComputation.prepare() re-entered while the set-up is running")
        self._preparing = True
        try:
            self._prepare(*self._args)
        except BaseException:
            # a failed set-up (size limit of the dense mode, hipMalloc, a singular factorisation) leaves the object as it was:
            # the real error reaches the caller and prepare() can be tried again (e.g. with another ASB_GEODESIC)
            for name in GeodesicDistanceComputation._PREPARED:
                self.__dict__.pop(name, None)
            raise
        finally:
            self._preparing = False
        self._ready = True
        self._args = None
        return self

    def _prepare(self, verts, tris, m, engine, tol, backend):
        n, M = verts.shape[0], tris.shape[0]
        p0, p1, p2 = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
        e01, e12, e20 = p1 - p0, p2 - p1, p0 - p2
        area = 0.5 * _vlen(np.cross(e01, e12))
        nrm = _unit(np.cross(_unit(e01), _unit(e12)))
        # gradient: grad u = 1/(2A) * sum_i u[v_i] * (n x e_opposite(i))   (:184-188)
        g_for = {2: np.cross(nrm, e01), 0: np.cross(nrm, e12), 1: np.cross(nrm, e20)}
        rows, cols, vals = [], [], []
        tri_ids = np.arange(M)
        for corner, vec in g_for.items():
            coef = vec / (2 * area)[:, None]
            for d in range(3):
                rows.append(3 * tri_ids + d)
                cols.append(tris[:, corner])
                vals.append(coef[:, d])
        self.G = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                   shape=(3 * M, n))
        # divergence: div[v1] += 0.5*(cot1*(e1.X) + cot2*(e2.X))               (:192-204)
        rows, cols, vals = [], [], []
        for a, b, c in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
            pa, pb, pc = verts[tris[:, a]], verts[tris[:, b]], verts[tris[:, c]]
            e1, e2, eo = pb - pa, pc - pa, pc - pb
            cot1 = 1 / np.tan(np.arccos((_unit(-e2) * _unit(-eo)).sum(axis=1)))
            cot2 = 1 / np.tan(np.arccos((_unit(-e1) * _unit(eo)).sum(axis=1)))
            coef = 0.5 * (cot1[:, None] * e1 + cot2[:, None] * e2)
            for d in range(3):
                rows.append(tris[:, a])
                cols.append(3 * tri_ids + d)
                vals.append(coef[:, d])
        self.D = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                   shape=(n, 3 * M))
        h = np.mean([_vlen(e01), _vlen(e12), _vlen(e20)])
        t = m * h ** 2
        L, vert_area = cotan_laplacian(verts, tris)
        self._A_heat = (sparse.diags(vert_area) - t * L).tocsr()
        self._L = L
        self._engine = None
        if engine is not None:          # device backend instead of the SuperLU factorisations
            coarse = None
            slabs = None
            if backend == "slab":                    # direct block-tridiagonal factorisation over breadth-first slabs
                slabs = bfs_slabs(self._A_heat.tocsr())
                self.n_slabs = int(slabs[1].shape[0] - 1)
                self.largest_slab = int(np.diff(slabs[1]).max())
            elif backend != "dense" and n >= 512:      # sparse mode: aggregates + dense coarse operators (two-level PCG)
                agg, nc = mesh_aggregates(self._A_heat.tocsr())
                Hc, Lc = coarse_operators(self._A_heat, L, agg, nc)
                # damping of the heat step's Jacobi sweeps: 1 on diagonally dominant matrices (non-obtuse meshes), below
                # where obtuse triangles (negative cotan weights) push lambda_max(D^-1 A) <= 1 + g towards 2 and beyond
                dg = self._A_heat.diagonal()
                g = float(((abs(self._A_heat).sum(axis=1).A1 - dg) / dg).max())
                omega = 1.0 if g <= 0.98 else min(1.0, 1.8 / (1.0 + g))
                coarse = (agg, Hc, Lc, omega)
                self.n_aggregates = nc
            engine.geodesic_setup(self._A_heat, (-L).tocsr(), self.G, self.D, dense=(backend == "dense"), coarse=coarse, slabs=slabs)
            self._engine = engine
            self._tol = tol
            return self
        # both matrices are symmetric: the minimum-degree ordering on A + A^T halves SuperLU's fill against the
        # default COLAMD (2.2M -> 1.05M non-zeros on a 15k-vertex mesh), i.e. 2-4x faster triangular solves; the
        # solutions agree with the default ordering (what the reference uses) to 4e-13
        kw = dict(permc_spec="MMD_AT_PLUS_A", options=dict(SymmetricMode=True))
        self._heat = splu(self._A_heat.tocsc(), **kw)
        try:
            self._poisson = splu(L.tocsc(), **kw)
        except RuntimeError:            # L is singular (constants): on some (tiny) meshes this ordering meets an exact
            self._poisson = splu(L.tocsc())     # zero pivot; the default ordering is what the reference factorises
        return self

    def _field(self, U):
        """U: (n,) or (n,k) heat solutions -> distances, same shape."""
        g = self.G @ U
        g3 = g.reshape((-1, 3) + g.shape[1:])
        Xf = -g3 / np.sqrt((g3 * g3).sum(axis=1, keepdims=True))
        phi = self._poisson.solve(np.ascontiguousarray(self.D @ Xf.reshape(g.shape)))
        return phi - phi.min(axis=0)

    def __call__(self, idx):
        return self.solve_many([int(idx)])[0].copy()

    def solve_many(self, idxs):
        """Distances from each vertex in ``idxs``: (len(idxs), n).  One multi-RHS solve for the vertices not asked
        for before: the field of a source never changes, and SPLOCS asks for the same centres again in every outer
        iteration once they have settled (posComponents.py:158-165), so solved fields are kept (<= ``cache_bytes``)."""
        self.prepare()
        idxs = np.asarray(idxs, dtype=np.int64)
        uniq = np.unique(idxs)
        new = np.array([i for i in uniq.tolist() if i not in self._cache], dtype=np.int64)
        if new.size:
            if self._engine is not None:
                parts, its = [], []
                for b in range(0, new.size, 64):
                    phi, it = self._engine.geodesic_solve(new[b:b + 64], self._tol)
                    parts.append(phi)
                    its.append(it)
                self.last_iterations = its
                fields = np.concatenate(parts, axis=0)
            else:
                E = np.zeros((self.n, new.size))
                E[new, np.arange(new.size)] = 1.0
                fields = np.ascontiguousarray(self._field(self._heat.solve(E)).T)
            room = max(int(self.cache_bytes // (8 * self.n)) - len(self._cache), 0)
            local = {}
            for q, i in enumerate(new.tolist()):
                (self._cache if q < room else local)[i] = fields[q]
        else:
            local = {}
        return np.stack([self._cache[i] if i in self._cache else local[i] for i in idxs.tolist()])
