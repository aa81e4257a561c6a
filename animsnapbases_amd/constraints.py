"""``nonlinearSnapshots`` / ``constraintsComponents`` -- mirror of the reference classes
(snapbases/nonlinear_snapshots.py, snapbases/constraintsComponents.py) for the part of them
that BASELINE config 5 names: constraint-projection snapshots -> standardise -> POD
(``pod_vectorized``) -> post-processing -> DEIM interpolation points -> ``.npz``.

Device work (csrc/asb_pod.hip): the (3ep x F) snapshot matrix stays in HBM in the same
row layout as the position path; its Gram matrix A^T A is accumulated with f64 MFMA over
the row shard (partial Gram matrices all-reduced over ranks with RCCL), the K leading
left singular vectors are formed by the deflation's projection kernel, QR per dimension is
CholeskyQR2, and each DEIM step's residual GEMV + arg-max runs on the shard.
The F x F symmetric eigen-problem, the K x K Cholesky factors and the small SVD of the Rayleigh-Ritz
step run on the device too (csrc/asb_eig.hip, csrc/asb_smalldense.hip).  Host work: DEIM's k x k
interpolation solves (an O(k^2) bordered-inverse update whose residual is checked every step;
``numpy.linalg.lstsq``, the routine the reference calls at constraintsComponents.py:829, is
the fallback); LAPACK ``eigh`` on the Gram matrix only with ``ASB_POD_EIG=host``.

Other basis types of the reference (``pod`` per-(p,d) torch SVD, ``pca_blocks``,
``pca_blocks_with_St``, geometric / block DEIM, polyscope views) are out of scope
 and raise ``NotImplementedError``.
"""
import csv
import os
import sys

import numpy as np

from . import _lib
from . import utils as _u
from ._panels import deflate_panels_multirank
from .distributed import Comm
from .engine import HipEngine
from .utils import log_time, summed_grams, testSparsity, test_linear_dependency, test_linear_dependency_grams


def read_sparse_matrix(file_name, file_type, key=None):
    """The sparse operator files of the reference (utils/utils.py:289-323): ``.npz`` holding a pickled scipy matrix under
    ``key``, or the C++ recorder's ``.bin`` -- ``<i rows><i cols><i nnz>`` then nnz triplets ``<i row><i col><d value>``."""
    from scipy.sparse import csr_matrix
    if file_type == ".npz":
        if key is None:
            raise ValueError("Empty file or wrong key:", file_type)
        M = np.load(file_name, allow_pickle=True)[key]
        return M.item() if isinstance(M, np.ndarray) and M.dtype == object else M
    if file_type == ".bin":
        raw = np.fromfile(file_name, dtype=np.uint8)
        rows, cols, nnz = np.frombuffer(raw[:12].tobytes(), dtype="<i4")
        trip = np.frombuffer(raw[12:12 + 16 * int(nnz)].tobytes(), dtype=np.dtype([("r", "<i4"), ("c", "<i4"), ("v", "<f8")]))
        return csr_matrix((trip["v"], (trip["r"], trip["c"])), shape=(int(rows), int(cols)))
    raise ValueError("unknown sparse matrix file_type:", file_type)


def elements_of_vertex(v, elems):
    """Indices (ascending) of the elements -- tets, triangles or edges, one per row of ``elems`` -- that contain vertex v:
    what get_tetrahedrons_per_vert / get_triangles_per_vert / get_edges_per_vert return for ``[v]``
    (utils/support.py:210-258), as one vectorised comparison."""
    return np.flatnonzero((np.asarray(elems) == v).any(axis=1)).tolist()


def vertex_star(v, faces):
    """get_vert_star_per_vert (utils/support.py:239-246): the vertices of every face that contains v, v itself included, in
    the order ``list(set)`` yields them -- reproduced by building the same set in the same insertion order."""
    faces = np.asarray(faces)
    star = set()
    for f in faces[(faces == v).any(axis=1)]:
        star.update(int(q) for q in f)
    return list(star)

constProj_output_directory = ""


class nonlinearSnapshots:
    """Constraint-projection snapshots: F frames of (e*p, 3) (nonlinear_snapshots.py:17-53)."""

    def __init__(self, param, *, frames=None, test_frames=None, engine=None, comm=None, mass=None, frames_device=None,
                 keepalive=None):
        self.snapshots_file = ""
        self.rest_shape = ""
        self.dim = 0
        self.mass_file = ""
        self.frs = 0
        self.constraintsSize = 0
        self.num_constained_elements = 0
        self.mean = None
        self.pre_scale_factor = 1
        self.mass = None
        self.massL = None
        self.invMassL = None
        self._snapTensor = None
        self.test_snapTensor = None
        # element lists of the simulation mesh (the reference fills them from mesh files through libigl; here they are handed
        # over, e.g. ns.tris = ...): used by the S^T variants to count / list the elements around a position-space vertex
        self.verts = self.tris = self.tets = self.edges = None
        self.ele_type = ""
        self.param = param
        self._frames, self._test_frames, self._preset_mass = frames, test_frames, mass
        # (dev_ptr, F, rows): an (F, rows, 3) float64 tensor already in this rank's HBM -- this rank's shard of the
        # constraint rows (or all of them); standardised in place like posSnapshots.from_device
        self._frames_device, self._keepalive = frames_device, keepalive
        self._engine, self._comm = engine, comm if comm is not None else Comm()
        self._shards = None

    # the prepared tensor lives on the device; (F, ep, 3) on read
    @property
    def snapTensor(self):
        if self._snapTensor is None and self._engine is not None and self._engine.n_loc:
            loc = self._engine.download_snapshots()
            self._snapTensor = self._comm.all_gather_rows(loc, self.frames_rows, axis=1)
        return self._snapTensor

    @snapTensor.setter
    def snapTensor(self, v):
        self._snapTensor = v

    def config(self):
        """nonlinear_snapshots.py:55-71."""
        global constProj_output_directory
        p = self.param
        self.snapshots_file = getattr(p, "constProj_input_snapshots_pattern", "")
        self.rest_shape = p.constProj_rest_shape
        self.dim = getattr(p, "constProj_dim", 3)
        self.mass_file = getattr(p, "constProj_masses_file", "")
        self.frs = p.constProj_numFrames
        self.constraintsSize = p.constProj_p_size
        self.ele_type = getattr(p, "constProj_element_type", "")
        constProj_output_directory = getattr(p, "constProj_output_directory", "")

    @log_time(constProj_output_directory)
    def snapshots_prepare(self):
        """nonlinear_snapshots.py:74-96: read, optional sqrt-mass weighting, standardise -- on the GPU."""
        X = self.read() if self._frames_device is None else None
        if self._engine is None:
            dev, stream = 0, None
            if self._comm.multi or self._frames_device is not None:
                import torch
                dev, stream = torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream
            self._engine = HipEngine(dev, stream)
        eng, comm = self._engine, self._comm
        if self._frames_device is not None:
            ptr_, F, rows = self._frames_device
            counts = comm.allreduce_sum(np.eye(comm.world)[comm.rank] * rows) if comm.multi else np.array([rows])
            self._shards, v0 = [], 0
            for n in counts.astype(np.int64):
                self._shards.append((v0, int(n)))
                v0 += int(n)
            self.frames_rows = int(counts.sum())
            self.frs = int(F)
            self.num_constained_elements = self.frames_rows // self.constraintsSize
            massL = None
            if self.param.constProj_massWeight:             # (:82-88: M^{1/2} X, applied in the layout-change sweep)
                self.load_factorize_masses()
                assert self.massL.shape[0] == self.frames_rows
                r0 = self._shards[comm.rank][0]
                massL = np.ascontiguousarray(self.massL[r0:r0 + int(rows)])
            fused = None
            if self.rest_shape in ("first", "average") and self.param.constProj_standarize:
                fused = eng.adopt_device_rest(int(ptr_), int(F), int(rows), massL, self._shards[comm.rank][0], self.frames_rows,
                                              0 if self.rest_shape == "first" else 1, True)
            else:
                eng.adopt_device(int(ptr_), int(F), int(rows), massL, self._shards[comm.rank][0], self.frames_rows)
            if self.param.constProj_standarize:
                self.standarize(_fused=fused)
            print('nonlinearSnapshots ready ... Volkwein (' + str(self.param.constProj_massWeight) + '), standarized (' +
                  str(self.param.constProj_standarize) + ').')
            return
        self.frames_rows = X.shape[1]
        massL = None
        if self.param.constProj_massWeight:
            self.load_factorize_masses()
            assert X.shape[1] == self.massL.shape[0]
            massL = self.massL
        v0, n_loc = comm.my_shard(X.shape[1])
        self._shards = comm.shards(X.shape[1])
        if min(n for _, n in self._shards) == 0:
            raise ValueError("%d constraint rows cannot be sharded over %d ranks: every rank needs at least one row"
                             % (X.shape[1], comm.world))
        fused = None
        if self.param.constProj_standarize and self.rest_shape in ("first", "average") and hasattr(eng, "upload_rest"):
            fused = eng.upload_rest(X, v0, n_loc, massL, 0 if self.rest_shape == "first" else 1, True)
        else:
            eng.upload(X, v0, n_loc, massL)
        if self.param.constProj_standarize:
            self.standarize(_fused=fused)
        print('nonlinearSnapshots ready ... Volkwein (' + str(self.param.constProj_massWeight) + '), standarized (' +
              str(self.param.constProj_standarize) + ').')

    @log_time(constProj_output_directory)
    def read(self, file_type=".npz"):
        """nonlinear_snapshots.py:99-173 ('.npz' keyed by the frame number; per-frame '.bin' files are the
        PD recorder's other format and out of scope)."""
        p = self.param
        if self._frames is not None:
            X = np.asarray(self._frames, dtype=np.float64)
            self.test_snapTensor = self._test_frames
        else:
            inc = p.constProj_frame_increment
            jump = getattr(p, "constProj_train_test_jump", None)
            if file_type == ".npz":
                data = np.load(self.snapshots_file, allow_pickle=True)
                frame = lambda i: data[str(i)]
            elif file_type == ".bin":
                # the PD recorder's per-frame files (:126-160): <pattern><i>.bin = <i rows><i cols> + rows x cols doubles,
                # column by column
                def frame(i):
                    raw = np.fromfile(self.snapshots_file + str(i) + ".bin", dtype=np.uint8)
                    ni, mi = (int(q) for q in np.frombuffer(raw[:8].tobytes(), dtype="<i4"))
                    return np.frombuffer(raw[8:8 + 8 * ni * mi].tobytes(), dtype="<f8").reshape(mi, ni).T
            else:
                raise ValueError("unknown snapshot file type: " + str(file_type))
            X = np.stack([frame(i) for i in range(0, self.frs * inc, inc)]).astype(np.float64)
            if jump:
                self.test_snapTensor = np.stack([frame(j) for j in range(jump, self.frs * inc, inc)])
        self.frs = X.shape[0]
        self.num_constained_elements = X.shape[1] // self.constraintsSize
        print("loaded snapshots size", X.shape)
        print("No. constrained verts: ", self.num_constained_elements)
        return X

    def load_factorize_masses(self):
        """nonlinear_snapshots.py:178-265.  The mass FILE branch (:180-191: the simulator's auxiliary masses, ``<i n><i m>`` +
        n doubles, one per constraint row) and the factorisation (:244-262: massL = sqrt(mass), invMassL = 1 / massL with 0
        where the mass is 0); a per-row vector can also be handed over directly (``mass=``; ``.npy``).  Without a file the
        masses come from the MESH (:192-240, ``_masses_from_mesh``)."""
        m = self._preset_mass
        if m is None and (not self.mass_file or not os.path.exists(self.mass_file)):
            m = self._masses_from_mesh()
        if m is None:
            if self.mass_file.lower().endswith(".npy"):
                m = np.load(self.mass_file)
            else:
                with open(self.mass_file, "rb") as fh:
                    ni, _mi = np.frombuffer(fh.read(8), dtype="<i4")
                    m = np.frombuffer(fh.read(8 * int(ni)), dtype="<f8")
                    if m.shape[0] != int(ni):
                        raise IOError(self.mass_file + " could not be read")
        self.mass = np.array(m, dtype=np.float64)
        massL = np.sqrt(self.mass)
        invMassL = np.zeros_like(massL)
        np.divide(1.0, massL, out=invMassL, where=massL != 0)
        assert np.allclose(invMassL * massL, np.ones_like(massL))             # (:258, as the reference: a zero mass fails here)
        self.massL, self.invMassL = massL, invMassL

    def _masses_from_mesh(self):
        """nonlinear_snapshots.py:192-240: per constraint row, the summed vertex masses of its element.  The element sums
        (utils/support.py:12-76) and the tetrahedral lumping of the volumetric p = 1 branch (``compute_lumped_mass_matrix``,
        :41-59) are the reference's own arithmetic, pinned by tests/golden/mesh_masses.npz; the VERTEX masses of the other
        branches come from libigl in the reference (``igl.massmatrix``: mixed Voronoi areas on triangles, a quarter of the
        volume per corner on tetrahedra) and are restated from their definitions here (no libigl in this image: "parity
        unpinned", checked against closed forms)."""
        p = self.param
        size = int(self.constraintsSize)
        tet_mesh, tri_mesh = getattr(p, "tet_mesh_file", None), getattr(p, "tri_mesh_file", None)
        n_el = int(self.num_constained_elements)
        if size == 1:
            if getattr(p, "volumetric_mesh", False):
                self.verts, self.tets, self.tris = _u.read_mesh_file(tet_mesh)
                vm = _u.lumped_tet_vertex_masses(self.verts, self.tets)                       # (:210)
            else:
                self.verts, self.tris = _u.read_triangle_mesh(tri_mesh)
                vm = _u.voronoi_vertex_masses(self.verts, self.tris)                          # (:213)
            kind = getattr(p, "constProj_snapshots_type", "")
            if kind == "verts_bending":
                verts = np.load(p.constProj_input_snaps_constrained_elements)["indices"]
                return vm[verts]
            if kind == "edge_spring":
                self.edges = _u.mesh_edges(self.tets if getattr(p, "volumetric_mesh", False) else self.tris)
                if self.edges.shape[0] != n_el:
                    raise ValueError("the mesh has %d edges, the snapshots %d constrained elements" % (self.edges.shape[0], n_el))
                return _u.element_masses(vm, self.edges, size)
            raise ValueError("masses from the mesh: unknown constProj_snapshots_type %r for p = 1" % (kind,))
        if size == 2:
            self.verts, self.tris = _u.read_triangle_mesh(tri_mesh)
            vm = _u.voronoi_vertex_masses(self.verts, self.tris)
            if self.tris.shape[0] != n_el:
                raise ValueError("the mesh has %d triangles, the snapshots %d constrained elements" % (self.tris.shape[0], n_el))
            return _u.element_masses(vm, self.tris, size)
        if size == 3:
            self.verts, self.tets, self.tris = _u.read_mesh_file(tet_mesh)
            vm = _u.tet_barycentric_vertex_masses(self.verts, self.tets)
            if self.tets.shape[0] != n_el:
                raise ValueError("the mesh has %d tetrahedra, the snapshots %d constrained elements" % (self.tets.shape[0], n_el))
            return _u.element_masses(vm, self.tets, size)
        raise ValueError("masses from the mesh: constraint size %d" % size)

    def standarize(self, _fused=None):
        """nonlinear_snapshots.py:268-288 (``_fused``: sum(x), sum(x^2) when the upload sweep already removed the rest
        shape -- see posSnapshots.standarize)."""
        eng, comm = self._engine, self._comm
        if self.rest_shape not in ("first", "average"):
            print('Error! unknown rest shape: ', self.rest_shape)
            sys.exit(1)
        local_sum, local_sumsq = _fused if _fused is not None else (eng.center(0 if self.rest_shape == "first" else 1, True), None)
        self.mean = comm.all_gather_rows(eng.get_mean(), self.frames_rows, axis=0)
        count = float(self.frs) * float(self.frames_rows) * 3.0
        var = None
        if local_sumsq is not None:
            tot = comm.allreduce_sum([local_sum, local_sumsq])
            mu = tot[0] / count
            v = tot[1] / count - mu * mu
            if v > 0 and mu * mu <= 10.0 * v:
                var = v
        else:
            mu = comm.allreduce_sum(local_sum)[0] / count
        if var is None:
            var = comm.allreduce_sum(eng.sqdev(mu))[0] / count
        self.pre_scale_factor = 1 / np.sqrt(var)
        eng.scale(self.pre_scale_factor)
        self._snapTensor = None


class constraintsComponents:  # Components == bases
    def __init__(self, param, nonlinear_snapshots=None):
        global constProj_output_directory
        constProj_output_directory = getattr(param, "constProj_output_directory", "")
        self.basesType = ""
        self.numComp = 0
        self.support = ""
        self.storeSingVal = False
        self.nonlinearSnapshots = nonlinear_snapshots if nonlinear_snapshots is not None else nonlinearSnapshots(param)
        self.param = param
        self._comps = None
        self._comps_on_device = False
        self.weigs = None
        self.fileNameBases = ""
        self.fileName_geom_points = ""
        self.file_name_sing = ""
        self.geom_interpol_verts = []
        self.geom_alpha = None
        self.geom_alpha_ranges = None
        self.geom_Pt = None
        self.St = None
        self.singular_values = None

    @property
    def comps(self):
        if self._comps is None and self._comps_on_device:
            ns = self.nonlinearSnapshots
            loc = ns._engine.results_comps()
            self._comps = ns._comm.all_gather_rows(loc, ns.frames_rows, axis=1)
        return self._comps

    @comps.setter
    def comps(self, v):
        self._comps = v
        self._comps_on_device = False

    def config(self, fileNameBases="p_nl_", fileName_geom_points="p_nl_interpol_points_",
               file_name_sing="_constrprojBases_pcaExtraction_singValues"):
        """constraintsComponents.py:61-74; the weighted S^T operator (``constProj_weightedSt``, an .npz holding a scipy
        sparse matrix under ``costProj_St_key``) is read when configured -- 'pca_blocks_with_St' and the position-space
        interpolation error use it (``self.St`` may also be assigned directly)."""
        p = self.param
        self.basesType = getattr(p, "constProj_bases_interpolation_type", "")
        self.support = getattr(p, "constProj_support", "global")
        self.storeSingVal = getattr(p, "constProj_store_sing_val", False)
        self.fileNameBases = fileNameBases
        self.fileName_geom_points = fileName_geom_points
        self.file_name_sing = file_name_sing
        st_file = getattr(p, "constProj_weightedSt", None)
        if st_file:
            self.St = read_sparse_matrix(st_file, ".npz", key=getattr(p, "costProj_St_key", None))

    def _elements_around(self, v):
        """The elements the reference lists around position-space vertex v (constraintsComponents.py:182-194, 682-697)."""
        ns = self.nonlinearSnapshots
        kind = getattr(ns, "ele_type", "")
        if kind == "_tets":
            return elements_of_vertex(v, ns.tets)
        if kind == "_tris":
            return elements_of_vertex(v, ns.tris)
        if kind == "_edges":
            return elements_of_vertex(v, ns.edges)
        if kind == "_verts":
            return vertex_star([v], ns.tris)
        raise ValueError("ERROR! unknown constraints projection type: %r" % (kind,))

    def _st_on_device(self):
        ns = self.nonlinearSnapshots
        if self.St is None:
            raise ValueError("the differential operator S^T is not set (constProj_weightedSt / costProj_St_key, or .St = ...)")
        if ns._comm.multi:
            raise NotImplementedError("the S^T variants run on one rank (every constraint row is needed for S^T R)")
        if self.St.shape[1] != ns.frames_rows:
            raise ValueError("S^T has %d columns, the snapshots %d constraint rows" % (self.St.shape[1], ns.frames_rows))
        if getattr(self, "_st_uploaded", None) is not self.St:
            ns._engine.st_upload(self.St)
            self._st_uploaded = self.St

    @log_time(constProj_output_directory)
    def compute_components_store_singvalues(self):
        """constraintsComponents.py:108-153."""
        p = self.param
        kind = p.constProj_basis_type
        if kind not in ("pod_vectorized", "pca_blocks", "pca_blocks_with_St", "pod"):
            raise ValueError("Uknown basis type: ", kind)
        if kind in ("pca_blocks", "pca_blocks_with_St"):
            headerSing = ['component', 'idx', 'residual_matrix_norm'] + \
                         ['singVal' + str(i) for i in range(self.nonlinearSnapshots.constraintsSize)]
            run = self.compute_nonlinearity_bases_blocks if kind == "pca_blocks" else \
                self.compute_nonlinearity_bases_blocks_utilizing_diffirential_operator
        elif kind == "pod":
            headerSing = ['component', 'singVal']
            run = self.compute_pod_for_nonlinear_snapshots_tensor
        else:
            headerSing = ['component', 'singVal']
            run = self.compute_pod_for_vectorized_nonlinear_snapshots_tensor
        file_name = os.path.join(p.constProj_output_directory, p.name + "_" + p.constProj_name + self.file_name_sing)
        rank0 = self.nonlinearSnapshots._comm.rank == 0
        if self.storeSingVal and rank0:
            with open(file_name + '.csv', 'w', encoding='UTF8') as singFile:
                writer = csv.writer(singFile)
                writer.writerow(headerSing)
                run(writer)
        else:
            run(None)

    @log_time(constProj_output_directory)
    def compute_nonlinearity_bases_blocks_utilizing_diffirential_operator(self, writer=None):
        """constraintsComponents.py:156-271 ('pca_blocks_with_St'), as the reference WRITES it: while |R| > bases_R_tol, the
        position-space vertex v with the largest row of S^T R is found (one SpMM + row reduction on the device), the
        elements around it are listed -- and then the loop ``for idx in range(len(elems))`` deflates the constraint blocks
        0 .. len(elems) - 1 (the loop INDEX, not the listed elements, is what the reference uses as block number, :201-207),
        p rows each, rank-1 SVD of the row's 3 x F slab, global support; CSV row and measures per block, early exit once
        |R| < tol.  The residual lives on the device (residual mode, forced rows); one rank."""
        if self.support == 'local':
            raise ValueError("Local support maps are not yet available for nonlinear-term components")
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        self._st_on_device()
        p = int(ns.constraintsSize)
        tol = float(self.param.bases_R_tol)
        # every deflation removes one direction of frame space from the whole residual: F of them empty it in exact
        # arithmetic; what a tolerance above rounding level needs is bounded by that plus one sweep over the blocks
        # (cap is only the loop bound: the basis and weight buffers start small and grow geometrically with the components
        # actually computed -- asb_deflate_reserve -- as the reference's lists do; cap components of e p x 3 doubles each
        # would be (e p)^2 x 24 bytes up front)
        cap = int(ns.frs) + p * int(ns.num_constained_elements) + p
        room = min(cap, max(4 * p, 64))
        eng.deflate_begin(room, False, _lib.DEFLATE_RESIDUAL)
        S_v_idx, S_ele_idns, meas = [], [], []
        n_done = bases_count = 0
        normR = np.sqrt(max(eng.residual_norm2(), 0.0))
        while normR > tol:
            v, _ = eng.st_residual_argmax()
            elems = self._elements_around(v)
            print("vert", v, "elements", len(elems))
            S_v_idx.append(v)
            if not elems:
                raise ArithmeticError("vertex %d has no element: the reference's loop would spin forever here" % v)
            for idx in range(len(elems)):
                if n_done + p > cap:
                    raise ArithmeticError("the residual does not fall below bases_R_tol = %g within %d components" % (tol, cap))
                if n_done + p > room:
                    room = min(cap, max(2 * room, n_done + p))
                    eng.deflate_reserve(room)
                S_ele_idns.append(idx)
                sigma = []
                for i in range(p):
                    eng.force_next(idx * p + i)
                    eng.pick(n_done)
                    eng.apply(n_done)
                    sigma.append(eng.get_pick(n_done)[1])
                    n_done += 1
                    print(np.sqrt(max(eng.residual_norm2(), 0.0)))
                bases_count += 1
                normR = np.sqrt(max(eng.residual_norm2(), 0.0))
                singList = [bases_count, idx, normR] + sigma
                meas.append(singList)
                if self.storeSingVal and writer is not None:
                    writer.writerow(singList)
                if normR < tol:
                    break
        for what, lst in (("verts", S_v_idx), ("elements", S_ele_idns)):
            if len(lst) == len(set(lst)):
                print("PCA Large deformation %s are unique%s" % (what, ":" if what == "elements" else ""), len(lst))
            else:
                print("PCA Large deformation %s are not unique:" % what, len(set(lst)), "points out of", len(lst))
        eng.components_truncate(n_done)
        res = eng.results(want_comps=False, want_weigs=True)
        self.weigs = res["weigs"]
        self.largeDeforPoints = np.asarray(S_v_idx, dtype=np.int64)
        self.largeDeforBlocks = np.asarray(S_ele_idns, dtype=np.int64)
        self.measures_at_largeDeforVerts = np.array(meas)
        self._comps, self._comps_on_device = None, True
        self.numComp = n_done // p
        print("bases shape", (n_done, ns.frames_rows, 3), "number of components", self.numComp)

    @log_time(constProj_output_directory)
    def compute_nonlinearity_bases_blocks(self, writer=None):
        """constraintsComponents.py:324-412 ('pca_blocks'): K times, the constraint whose p rows carry the most residual
        energy is chosen and its p rows are deflated one after the other (rank-1 SVD of the row's 3 x F slab, global
        support) -- the greedy deflation of posComponents on the constraint rows.  p = 1 IS that loop (the block
        arg-max is the row arg-max), so it runs through the panel algorithm; p > 1 keeps the residual on the device
        and names the row of every pick (asb_deflate_block_argmax / asb_deflate_force_next)."""
        if self.support == 'local':
            raise ValueError(' Local support is not yet available for nonlinearity')
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        p = int(ns.constraintsSize)
        K = int(self.param.deim_desired_num_components)
        Kp = K * p
        if p == 1:
            from .posComponents import _auto_global_mode
            small = _auto_global_mode(ns.frs, ns._shards[comm.rank][1], comm.multi) == "residual"
            eng.deflate_begin(Kp, False, _lib.DEFLATE_RESIDUAL if small else _lib.DEFLATE_PROJECT)
            if comm.multi:
                deflate_panels_multirank(eng, comm, ns.frames_rows, Kp)
            else:
                eng.run_global(0, Kp)
        else:
            for v0, n_loc in ns._shards:
                if v0 % p or n_loc % p:
                    raise ValueError("'pca_blocks' with p = %d needs shards of whole constraints; %d rows over %d ranks do "
                                     "not split that way" % (p, ns.frames_rows, comm.world))
            eng.deflate_begin(Kp, False, _lib.DEFLATE_RESIDUAL)
            rec = recs = None
            if comm.multi:
                rec, recs = comm.new_records(eng.xchg_len(), eng.device_exchange)
            for k in range(K):
                b, val = eng.block_argmax(p)
                if comm.multi:
                    b = int(comm.global_argmax(np.array([b]), np.array([val]))[0])
                for i in range(p):
                    eng.force_next(b * p + i)
                    if comm.multi:
                        eng.local_best(k * p + i, rec.data_ptr())
                        comm.all_gather_records(rec, recs)
                        eng.pick(k * p + i, recs.data_ptr(), comm.world)
                    else:
                        eng.pick(k * p + i)
                    eng.apply(k * p + i)
        res = eng.results(want_comps=False, want_weigs=True)
        normR = np.sqrt(np.maximum(comm.allreduce_sum(res["normR2_local"]), 0.0))     # |X|^2 - sum can round below 0 once |R| < 1e-8 |X|
        self.weigs = res["weigs"]
        self.largeDeforBlocks = np.asarray(res["idx"], dtype=np.int64)               # (K p,) rows, 0 <= . < e p
        self.largeDeforPoints = self.largeDeforBlocks[::p] // p                       # (K,) constraints
        meas = np.empty((K, 3 + p))
        meas[:, 0] = np.arange(K)
        meas[:, 1] = self.largeDeforPoints
        meas[:, 2] = normR[p - 1::p]                  # ||R|| after the constraint's last row
        meas[:, 3:] = res["sigma"].reshape(K, p)
        self.measures_at_largeDeforVerts = meas
        if writer is not None:
            for row in meas:
                writer.writerow([int(row[0]), int(row[1])] + [float(x) for x in row[2:]])
        self._comps, self._comps_on_device = None, True
        self.numComp = K
        print("bases shape", (Kp, ns.frames_rows, 3), "number of components", self.numComp)

    @log_time(constProj_output_directory)
    def compute_pod_for_nonlinear_snapshots_tensor(self, writer=None):
        """constraintsComponents.py:274-294 ('pod'): a batched SVD over the (p, d) slices of the snapshots, each an e x F
        matrix; component k carries the k-th left singular vector of every slice.  The reference runs torch's float32 SVD
        on the CPU; here every slice goes through Gram matrix -> device eigen-solver -> U = M V S^-1 in float64 (nothing is
        written to the CSV by the reference either)."""
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        if comm.multi:
            raise NotImplementedError("constProj_basis_type 'pod' runs on one rank")
        p = int(ns.constraintsSize)
        e = ns.frames_rows // p
        K = min(int(self.param.deim_desired_num_components), min(e, ns.frs))
        eng.pod_slices(p, K)
        self._comps, self._comps_on_device = None, True
        self.numComp = K

    @log_time(constProj_output_directory)
    def compute_pod_for_vectorized_nonlinear_snapshots_tensor(self, writer=None):
        """constraintsComponents.py:298-320: svd(A), A = R.reshape(F,-1).T (3ep x F): ``comps = U[:K]`` and ALL F singular values
        (the CSV).  Here: Gram matrix -> F x F eigen-problem on the device -> Rayleigh-Ritz on A itself (+ steps of subspace
        iteration for weak vectors) -- and, where K reaches below what the Gram matrix of A resolves (~1e-8 sigma_0; the
        reference's gesdd returns vectors there too), the same again on the DEFLATED snapshots A - U_1 U_1^T A, level by level
        (round 4; round 3 raised ArithmeticError)."""
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        F = ns.frs
        K = min(int(self.param.deim_desired_num_components), F)
        on_dev = os.environ.get("ASB_POD_EIG", getattr(self, "pod_eig", "device")) == "device"
        levels_ok = hasattr(eng, "pod_deflate_begin") and not (F & 1) and not any((3 * n) & 1 for _, n in ns._shards) \
            and os.environ.get("ASB_POD_LEVELS", "1") != "0"
        S_kept = []                     # singular values of the vectors kept by finished levels
        remaining, level = K, 0
        self.pod_power_steps = 0
        while True:
            S_lvl, Sb, Kx, Bbuf, keep = self._pod_level(remaining, on_dev)
            resolved = int(np.sum(S_lvl > 3e-8 * S_lvl[0])) if S_lvl[0] > 0 else 0
            if Kx >= remaining:         # this level delivers everything that is still wanted
                break
            # K reaches below what this level's Gram matrix resolves: keep what it resolved WELL (above 1e-5 of the level's
            # largest: Gram error below 1e-6, which the subspace-iteration steps remove), deflate, go on
            if not levels_ok or keep < 2 or level >= 8:
                raise ArithmeticError("POD: singular value %d of the %d requested is %.3e of the largest -- below what the "
                                      "Gram-matrix route resolves (1e-8), and the deflated levels cannot take over here (%s)"
                                      % (K, K, S_lvl[remaining - 1] / S_lvl[0] if S_lvl[0] > 0 else 0.0,
                                         "odd F or row count" if not levels_ok else "nothing left to resolve: the snapshot "
                                         "matrix has numerical rank %d" % (K - remaining + resolved)))
            eng.pod_deflate_begin(keep, Bbuf.data_ptr() if Bbuf is not None else None)
            S_kept.extend(Sb[:keep].tolist())
            remaining -= keep
            level += 1
        if level:
            eng.pod_deflate_end(remaining, K)
            # the levels' vectors are orthogonal to one another to ~eps |A| / |A_level|: one CholeskyQR2 over all K rows
            for _ in range(2):
                Gq = comm.new_buffer(3 * K * K, eng.device_exchange) if comm.multi else None
                eng.orth_gram(Gq.data_ptr() if Gq is not None else None)
                if Gq is not None:
                    comm.allreduce_tensor(Gq)
                eng.qr_apply_joint(Gq.data_ptr() if Gq is not None else None)
        elif Kx > K:                                    # drop the oversampling vectors again
            eng.components_truncate(K)
        S = np.concatenate([np.asarray(S_kept), S_lvl])[:F] if S_kept else S_lvl
        if Sb is not None:
            S = S.copy()
            nb = min(len(Sb), S.shape[0] - len(S_kept))
            S[len(S_kept):len(S_kept) + nb] = Sb[:nb]
        self.singular_values = S
        self.pod_levels = level + 1
        if writer is not None:
            for ai, bi in zip(range(1, S.shape[0] + 1), S):
                writer.writerow([ai, bi])
        self._comps, self._comps_on_device = None, True
        self.numComp = K

    def _pod_level(self, K, on_dev):
        """One level of the POD on the context's CURRENT snapshots (the original ones, or a deflated copy): all F singular
        values the Gram matrix gives, and a device basis of Kx rows -- the K wanted ones (+ oversampling while more follow) when
        the level resolves them, else as many as it does resolve.  Returns (S_level (F), refined values Sb (Kx) or None, Kx, B)."""
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        F = ns.frs
        # eigenvectors computed: K + the refinement's oversampling -- all F of them when that is not much more: a noise
        # floor is one big cluster of singular values, and a Rayleigh-Ritz subspace that cuts through the cluster leaves
        # the requested vectors inside it 1e-4 off (seed 41014 of tools/fuzz_sweep_more.py: F = 69, K = 22, rank 5 + noise)
        Kv = F if F <= 2 * (K + 32) else K + 32
        Gbuf = None
        if comm.multi:
            Gbuf = comm.new_buffer(F * F, eng.device_exchange)
            eng.pod_gram(Gbuf.data_ptr(), to_host=False)
            comm.allreduce_tensor(Gbuf)                      # partial Gram matrices: RCCL all-reduce
        dev_vectors = False
        if on_dev and F >= 3 and (Gbuf is None or Gbuf.is_cuda):
            # F x F eigen-problem entirely on the device: Householder tridiagonalisation (asb_eig.hip), bisection + inverse
            # iteration on the tridiagonal matrix (asb_smalldense.hip), back-transformation; ordered reductions only, so
            # every rank gets identical values and vectors from its copy of the all-reduced Gram matrix
            if Gbuf is None:
                eng.pod_gram(to_host=False)
            lam, n_bad = eng.sym_eig_topk(F, Kv, Gbuf.data_ptr() if Gbuf is not None else None)
            if n_bad:
                print("[asb] POD: %d of %d inverse iterations missed the growth criterion" % (n_bad, Kv))
            dev_vectors, V = True, None
        else:       # ASB_POD_EIG=host (explicit) or F < 3: LAPACK on the F x F Gram matrix
            G = Gbuf.cpu().numpy().reshape(F, F) if Gbuf is not None else eng.pod_gram()
            G = 0.5 * (G + G.T)
            lam, V = np.linalg.eigh(G)                       # ascending
            lam, V = lam[::-1], V[:, ::-1]
        S = np.sqrt(np.maximum(lam, 0.0))
        # The Gram route resolves singular values down to ~sqrt(eps) sigma_max (vector k carries an error eps (sigma_0 /
        # sigma_k)^2).  Below that a left vector A v / sigma is noise divided by noise: this level stops there.
        resolved = int(np.sum(S > 3e-8 * S[0])) if S[0] > 0 else 0
        if resolved < K:
            Kx = min(Kv, resolved)
            if Kx < 1:
                return S, None, 0, None, 0
        else:
            Kx = None
        # Gram-route accuracy of left vector k is eps (sigma_0 / sigma_k)^2, and the device's inverse iteration gives the
        # vectors of close eigenvalues only as a span.  Rayleigh-Ritz on A itself repairs both: K + p Gram vectors ->
        # orthonormal Q (CholeskyQR2 with the joint Gram matrix over all 3 ep entries, blocked Cholesky on the device) ->
        # B = Q^T A (one more pass over A) -> left vectors / singular values of the small B by one-sided Jacobi on the
        # device -> basis = Q U_B.  Always on with the device eigen-solver; with host LAPACK vectors only when the
        # weakest requested component makes the Gram route worse than ~1e-9.
        refine = (dev_vectors or Kx is not None or S[0] > 3e3 * S[K - 1]) and getattr(self, "pod_refine", True)
        if not refine:
            if dev_vectors:
                eng.pod_basis_dev(K)
            else:
                eng.pod_basis(np.ascontiguousarray(V[:, :K]), S[:K])
            return S, None, K, None, K
        if Kx is None:
            Kx = int(min(Kv, resolved))
        if dev_vectors:
            eng.pod_basis_dev(Kx)
        else:
            eng.pod_basis(np.ascontiguousarray(V[:, :Kx]), S[:Kx])

        def ritz():
            for _ in range(2):
                Gq = comm.new_buffer(3 * Kx * Kx, eng.device_exchange) if comm.multi else None
                eng.orth_gram(Gq.data_ptr() if Gq is not None else None)
                if Gq is not None:
                    comm.allreduce_tensor(Gq)
                eng.qr_apply_joint(Gq.data_ptr() if Gq is not None else None)       # one factor for all three slices
            Bbuf = comm.new_buffer(Kx * F, eng.device_exchange) if comm.multi else None
            eng.pod_project(Bbuf.data_ptr() if Bbuf is not None else None, to_host=False)
            if Bbuf is not None:
                comm.allreduce_tensor(Bbuf)
            return eng.pod_rotate(Bbuf.data_ptr() if Bbuf is not None else None), Bbuf
        Sb, Bbuf = ritz()
        # Tail accuracy (round 4).  Rayleigh-Ritz repairs what lies INSIDE span(Q); the Gram route also leaves every weak
        # vector eps (sigma_0 / sigma_k)^2 OUTSIDE it.  One step of subspace iteration -- basis <- A V Sigma^-1 from the
        # right Ritz vectors, then Rayleigh-Ritz again -- shrinks the part along a missing direction j by (sigma_j /
        # sigma_k)^2.  Taken where it is needed (the weakest wanted vector below 1e-4 sigma_0: Gram error above 1e-8; a second
        # step below 1e-5 -- config 5's fixture, sigma_256 = 3e-6 sigma_0: largest per-vector error 5e-6 without, 2.7e-7 with
        # one step, 1e-8 with two) AND can work (the spectrum still falls across the oversampling vectors; inside a flat noise
        # floor every missing direction is as strong as the vector itself and the step would change nothing).
        want_power = getattr(self, "pod_power", os.environ.get("ASB_POD_POWER", "1") != "0")
        # what this level hands on: the K wanted vectors, or -- when it cannot resolve them all -- the ones above 1e-5 of its
        # largest singular value (an even number: the device GEMMs' alignment)
        target = K
        if Kx < K:
            target = int(np.sum(Sb > 1e-5 * Sb[0]))
            target -= target & 1
        kq = max(target, 1) - 1
        steps = 0
        while want_power and hasattr(eng, "pod_power") and target >= 1 and Kx > target and Sb[Kx - 1] < 0.6 * Sb[kq] \
                and not ((Kx | F) & 1) and not any((3 * n) & 1 for _, n in ns._shards) \
                and steps < (2 if Sb[kq] < 1e-5 * Sb[0] else (1 if Sb[kq] < 1e-4 * Sb[0] else 0)):
            eng.pod_power(Bbuf.data_ptr() if Bbuf is not None else None)
            Sb, Bbuf = ritz()
            steps += 1
        self.pod_power_steps += steps
        return S, Sb, Kx, Bbuf, target

    @log_time(constProj_output_directory)
    def post_process_components(self):
        """constraintsComponents.py:415-446."""
        p = self.param
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        v0, n_loc = ns._shards[comm.rank]
        if not self._comps_on_device:
            eng.components_upload(np.ascontiguousarray(self._comps[:, v0:v0 + n_loc, :]))
            self._comps_on_device = True
        if p.constProj_standarize:
            eng.components_post(True, ns.pre_scale_factor, None, download=False)
            eng.snapshots_affine(1.0 / ns.pre_scale_factor, True, None)        # also restore the snapshots (:424-428)
            ns._snapTensor = None
        if p.constProj_orthogonal:
            for _ in range(2):                                                 # CholeskyQR2, any K, on the device
                Gbuf = comm.new_buffer(3 * self.numComp * self.numComp, eng.device_exchange) if comm.multi else None
                eng.orth_gram(Gbuf.data_ptr() if Gbuf is not None else None)
                if Gbuf is not None:
                    comm.allreduce_tensor(Gbuf)
                eng.qr_apply(Gbuf.data_ptr() if Gbuf is not None else None)
        if p.constProj_massWeight:
            assert ns.frames_rows == ns.invMassL.shape[0]
            eng.components_post(False, 1.0, ns.invMassL[v0:v0 + n_loc], download=False)
            eng.snapshots_affine(1.0, False, ns.invMassL[v0:v0 + n_loc])
            ns._snapTensor = None
        self._comps = None
        print("Post-processing, Undo standardization: ", p.constProj_standarize, ". Orthogonal-ized",
              p.constProj_orthogonal, ". Mass weighting", p.constProj_massWeight, ", and bases shape",
              (self.numComp, ns.frames_rows, 3))

    def deim(self):
        """constraintsComponents.py:797-860.  Residual GEMV + arg-max on the GPU shard(s); the k x k
        interpolation solves on the host (bordered inverse, verified per step, ``np.linalg.lstsq`` -- the
        reference's call at :829 -- as fallback)."""
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        p_size = ns.constraintsSize
        K = self.numComp
        on_device = not comm.multi and self._comps_on_device and hasattr(eng, "deim_run") and \
            os.environ.get("ASB_DEIM", getattr(self, "deim_backend", "device")) == "device"
        if not (on_device and hasattr(eng, "orth_gram_get")):
            self._rank_diagnostic(K)
        if on_device:
            # single rank: the whole loop on the device (bordered inverse of the k x k systems carried there, verified
            # per step); one host synchronisation.  A failed verification falls through to the lstsq loop below.
            # The printed rank check (:801) needs the eigenvalues of three K x K Gram matrices -- 7 ms of LAPACK on the host at
            # K = 256: the device loop is started on a worker thread (one C call, the interpreter lock released for all of it) and
            # the eigenvalues are computed here meanwhile; the check's line is printed first, as in the reference.
            if hasattr(eng, "orth_gram_get"):
                import threading
                G = summed_grams(eng, comm, K)
                box = {}

                def _loop():
                    try:
                        box["out"] = eng.deim_run()
                    except BaseException as exc:      # (re-raised on the calling thread)
                        box["exc"] = exc
                th = threading.Thread(target=_loop)
                th.start()
                lams = [np.linalg.eigvalsh(G[j]) for j in range(3)]
                th.join()
                if "exc" in box:
                    raise box["exc"]
                Pt_d, maxabs, bad = box["out"]
                test_linear_dependency_grams(G, K, lambda j: self.comps[:, :, j].T, lams=lams)
            else:
                Pt_d, maxabs, bad = eng.deim_run()
            if bad and os.environ.get("ASB_DEBUG_DEIM"):
                print("[asb] DEIM: the device loop's verification failed at a step: host loop", file=sys.stderr)
            if not bad:
                for k in range(K):
                    if k > 0 and maxabs[k] <= 1e-8:             # np.allclose(r, 0) of :837
                        print("ERROR!: zero residual!!")
                        return
                    print(k, int(Pt_d[k]) // p_size)
                self._set_interpolation(list(Pt_d), [int(i) // p_size for i in Pt_d], list(range(1, K + 1)), list(Pt_d))
                print("Regular Deim interpolation, used", self.geom_alpha.shape[0], "constrained elements")
                return
        rows = np.zeros((K, K, 3))          # rows[m, j, i] = V[Pt[m], j, i]
        Pt, e_points, e_range = [], [], []
        # The reference solves the growing k x k system from scratch with lstsq at every step (O(K^4) in all).
        # Here the inverse of V[Pt,:k,i] is carried along by the bordering (Schur-complement) update, O(k^2)
        # per step, and checked: if the solve's residual is not at rounding level the step falls back to lstsq.
        Minv = [np.zeros((0, 0)) for _ in range(3)]
        fro2 = np.zeros((3, K))
        for k in range(K):
            coef = None
            if k > 0:
                coef = np.empty((3, k))
                for i in range(3):
                    M, b = rows[:k, :k, i], rows[:k, k, i]
                    if Minv[i].shape[0] == k - 1:          # grow the inverse by the point / vector added last step
                        if k == 1:
                            Minv[i] = np.array([[1.0 / M[0, 0]]]) if M[0, 0] != 0 else None
                        else:
                            A_inv = Minv[i]
                            bcol, crow, d = M[:k - 1, k - 1], M[k - 1, :k - 1], M[k - 1, k - 1]
                            u = A_inv @ bcol
                            sch = d - crow @ u
                            if sch != 0 and np.isfinite(sch):
                                w = crow @ A_inv
                                new = np.empty((k, k))
                                new[:k - 1, :k - 1] = A_inv + np.outer(u, w) / sch
                                new[:k - 1, k - 1] = -u / sch
                                new[k - 1, :k - 1] = -w / sch
                                new[k - 1, k - 1] = 1.0 / sch
                                Minv[i] = new
                            else:
                                Minv[i] = None
                    x = Minv[i] @ b if (Minv[i] is not None and Minv[i].shape[0] == k) else None
                    if x is None or not np.all(np.isfinite(x)) or \
                            np.linalg.norm(M @ x - b) > 1e-10 * (np.linalg.norm(b) + np.sqrt(fro2[i, k - 1]) * np.linalg.norm(x)):
                        x = np.linalg.lstsq(M, b, rcond=None)[0]             # the reference's call (:829)
                        try:
                            Minv[i] = np.linalg.inv(M)
                        except np.linalg.LinAlgError:
                            Minv[i] = np.linalg.pinv(M)
                    coef[i] = x
            idx, val = eng.deim_step(k, coef)
            if comm.multi:
                idx = int(comm.global_argmax(np.array([idx]), np.array([val]))[0])
                val = float(comm.allreduce_max(val)[0])
            if k > 0 and not val > 1e-16:                # (all |r| <= 1e-8, np.allclose(r, 0) of :837, implies this)
                print("ERROR!: zero residual!!")
                return
            row = eng.deim_row(idx)
            if comm.multi:
                row = comm.allreduce_sum(np.zeros((K, 3)) if row is None else row).reshape(K, 3)
            rows[k] = row
            # running ||V[Pt[:m], :m, i]||_F^2 for the residual test above: the leading (k+1) x (k+1) block gains
            # row k (columns <= k) and column k (rows < k)
            for i in range(3):
                fro2[i, k] = (fro2[i, k - 1] if k else 0.0) + np.sum(rows[k, :k + 1, i] ** 2) + np.sum(rows[:k, k, i] ** 2)
            alpha = idx // p_size
            print(k, alpha)
            Pt.append(idx)
            e_points.append(alpha)
            e_range.append(k + 1)
        self._set_interpolation(Pt, e_points, e_range, Pt)
        print("Regular Deim interpolation, used", self.geom_alpha.shape[0], "constrained elements")

    def _set_interpolation(self, Pt, e_points, e_range, idxs=None):
        self.geom_Pt = np.array(Pt)
        self.geom_alpha = np.array(e_points)
        if idxs is not None:
            self.geom_ep_idxs = np.array(set(int(i) for i in idxs))      # (0-d object array, as the reference builds it :853)
        self.geom_alpha_ranges = np.array(e_range)
        self.geom_interpol_verts = np.array(self.geom_interpol_verts)

    # ------------------------------------------------------------------ diagnostics (host NumPy on the downloaded basis)
    def is_utmu_orthogonal(self):
        """constraintsComponents.py:452-461: prints True.. / False.. per coordinate for U^T M U = I."""
        print('... testing M orthogonality, U^T M U = I (Kp x Kp) ...', end='', flush=True)
        comps, mass = self.comps, self.nonlinearSnapshots.mass
        for l in range(comps.shape[2]):
            Mu_l = comps[:, :, l].T * mass[:, None]
            utMu_l = np.dot(comps[:, :, l], Mu_l)
            print('True..' if np.allclose(utMu_l, np.eye(comps.shape[0])) else "False..")

    def matrix_properties_test(self, interpol_kp_blocks, precondition=False):
        """constraintsComponents.py:463-487: per frame, coordinate and number of interpolation points i, the relative error of
        reconstructing the frame from its values at the first (i + 1) p interpolation rows (LU of the square J V).  Returns
        mat_e (F, points, 3).  The per-frame loop of the reference is one multi-right-hand-side solve here."""
        from scipy.linalg import lu_factor, lu_solve
        ns = self.nonlinearSnapshots
        p = ns.constraintsSize
        interpol_kp_blocks = np.asarray(interpol_kp_blocks)
        num_interpol_points = interpol_kp_blocks.shape[0] // p
        snap = ns.snapTensor
        F = snap.shape[0]
        bases = self.comps.swapaxes(0, 1)                       # (ep, Kp, d)
        denom = ns.dim * F * num_interpol_points * p
        mat_e = np.zeros((F, num_interpol_points, 3))
        for l in range(ns.dim):
            fn = np.linalg.norm(snap[:, :, l], axis=1)
            for i in range(num_interpol_points):
                points = interpol_kp_blocks[:(i + 1) * p]
                JV = bases[points, :(i + 1) * p, l]
                x = lu_solve(lu_factor(JV), snap[:, points, l].T)            # ((i + 1) p, F)
                r = bases[:, :(i + 1) * p, l] @ x - snap[:, :, l].T
                mat_e[:, i, l] = np.linalg.norm(r, axis=0) / denom / fn
        return mat_e

    @staticmethod
    def frobenius_error(f, f_reconstructed):
        """constraintsComponents.py:524-530."""
        return np.linalg.norm(f - f_reconstructed)

    @staticmethod
    def relative_error_per_component(f, f_reconstructed):
        """constraintsComponents.py:532-545."""
        return [np.linalg.norm(f[:, :, i] - f_reconstructed[:, :, i]) / np.linalg.norm(f[:, :, i]) for i in range(3)]

    @staticmethod
    def max_pointwise_error(f, f_reconstructed):
        """constraintsComponents.py:547-556."""
        return np.max(np.abs(f - f_reconstructed)) / np.max(f)

    def test_basesSingVals(self):
        """constraintsComponents.py:558-570: per-coordinate singular values of the (Kp x ep) basis, normalised."""
        from scipy.linalg import svd
        bases = self.comps.copy()
        s = np.empty((bases.shape[0], 3))
        for i in range(3):
            sing = svd(bases[:, :, i], full_matrices=False, compute_uv=False)
            s[:, i] = sing / sing.max()
        return s

    def _rank_diagnostic(self, expected):
        """test_linear_dependency(bases, 3, expected) of :801 / :630 / :742 -- from the device's Gram matrices when the
        basis lives there."""
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        if self._comps_on_device and hasattr(eng, "orth_gram_get"):
            test_linear_dependency_grams(summed_grams(eng, comm, expected), expected, lambda j: self.comps[:, :, j].T)
        else:
            test_linear_dependency(self.comps.swapaxes(0, 1), 3, expected)

    def _block_interpolation(self, group, unique):
        """Shared loop of deim_blocksForm (:733-795, group = 1: arg-max over rows) and of
        geom_block_form_utilizing_differential_operator in the constraint space (:619-731, group = p: arg-max over
        constraints, no constraint twice).  Step k: lstsq of V[Pt, :kp, i] against the block's p vectors (the reference's
        call, host; k p x k p), residual + arg-max on the device."""
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        if comm.multi:
            raise NotImplementedError("block interpolation runs on one rank")
        p = int(ns.constraintsSize)
        K = self.numComp
        Kp = K * p
        v0, n_loc = ns._shards[comm.rank]
        if not self._comps_on_device:
            eng.components_upload(np.ascontiguousarray(self._comps[:, v0:v0 + n_loc, :]))
            self._comps_on_device = True
        if eng.K != Kp:
            raise ValueError("the basis has %d vectors, %d blocks of %d expected" % (eng.K, K, p))
        self._rank_diagnostic(Kp)
        rows = np.zeros((0, Kp, 3))          # rows[m, j, i] = V[Pt[m], j, i]
        Pt, e_points, e_range, idxs = [], [], [], []
        for k in range(K):
            coef = None
            if k > 0:
                kp = k * p
                coef = np.empty((3, kp, p))
                for i in range(3):
                    coef[i] = np.linalg.lstsq(rows[:, :kp, i], rows[:, kp:kp + p, i], rcond=None)[0]      # (:764-765)
            idx, val, amax = eng.deim_block_step(k, p, coef, group)
            if k > 0 and amax <= 1e-8:                       # np.allclose(r, 0) (:768 / :677)
                print("ERROR!: zero residual!!")
                return False
            alpha = idx // p if group == 1 else idx
            if unique:
                assert alpha not in e_points
            idxs.append(idx)
            e_points.append(alpha)
            print(k, alpha)
            new = np.stack([eng.deim_row(alpha * p + m) for m in range(p)])
            rows = np.concatenate([rows, new], axis=0)
            Pt.extend(alpha * p + m for m in range(p))
            e_range.append(k + 1)
        self._set_interpolation(Pt, e_points, e_range, idxs if group == 1 else None)
        return True

    def deim_blocksForm(self):
        """constraintsComponents.py:733-795."""
        if self._block_interpolation(1, False):
            print("Regular Deim interpolation, used", self.geom_alpha.shape[0], "constrained elements")

    @log_time(constProj_output_directory)
    def geom_block_form_utilizing_differential_operator(self, error_in_pos_space=False):
        """constraintsComponents.py:619-731.  Constraint space (default): the block with the largest residual, no block
        twice.  ``error_in_pos_space=True``: the residual of the block is mapped to position space by S^T (device SpMM), the
        vertex with the largest row is the interpolation vertex, and of the elements around it those not taken yet -- at
        most ``geom_ele_per_vert`` per step -- join the interpolation set with all p rows (``verts_bending``: the listed
        vertices are first intersected with the constrained ones and enter by their position in that list, :692-696)."""
        if not error_in_pos_space:
            if self._block_interpolation(int(self.nonlinearSnapshots.constraintsSize), True):
                print("Computing interpolation elements utilizing differential operator, used", self.geom_alpha.shape[0],
                      "constrained elements")
            return
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        if getattr(ns, "ele_type", "") not in ["_tets", "_tris", "_edges", "_verts"]:
            print("ERROR! Unknown constained elements nonliner snapshots type!")
            return
        self._st_on_device()
        bending = getattr(self.param, "constProj_snapshots_type", "") == "verts_bending"
        if bending:
            self.constrianed_verts = np.load(self.param.constProj_input_snaps_constrained_elements)["indices"]
        p = int(ns.constraintsSize)
        K = self.numComp
        Kp = K * p
        v0, n_loc = ns._shards[comm.rank]
        if not self._comps_on_device:
            eng.components_upload(np.ascontiguousarray(self._comps[:, v0:v0 + n_loc, :]))
            self._comps_on_device = True
        if eng.K != Kp:
            raise ValueError("the basis has %d vectors, %d blocks of %d expected" % (eng.K, K, p))
        self._rank_diagnostic(Kp)
        per_vert = int(self.param.geom_ele_per_vert)
        rows = np.zeros((0, Kp, 3))          # rows[m, j, i] = V[Pt[m], j, i]
        Pt, e_points, e_jump, e_range = [], [], [], []
        self.geom_interpol_verts = []
        for k in range(K):
            coef = None
            if k > 0:
                kp = k * p
                coef = np.empty((3, kp, p))
                for i in range(3):          # (:662-668: more rows than columns possible -- least squares, as the reference)
                    coef[i] = np.linalg.lstsq(rows[:, :kp, i], rows[:, kp:kp + p, i], rcond=None)[0]
            v_interpolate, val, am = eng.deim_block_step_st(k, p, coef)
            if k > 0 and am <= 1e-8:     # np.allclose(S^T r, 0) (:677): no ENTRY of S^T r above 1e-8 (am = the largest |entry|)
                print("ERROR!: zero residual!!")
                return
            self.geom_interpol_verts.append(v_interpolate)
            alpha_list = self._elements_around(v_interpolate) if ns.ele_type != "_edges" else elements_of_vertex(v_interpolate, ns.edges)
            mapped = None
            if bending and ns.ele_type == "_verts":
                alpha_list, mapped, _ = np.intersect1d(self.constrianed_verts, alpha_list, return_indices=True)
            jump = 0
            new_rows = []
            for al in range(len(alpha_list)):
                alpha = int(alpha_list[al])
                if alpha not in e_points and jump < per_vert:
                    jump += 1
                    e_points.append(alpha)
                    if bending:
                        new_rows.append(int(mapped[al]))      # (p == 1 in this case)
                    else:
                        print(k, alpha)
                        new_rows.extend(alpha * p + m for m in range(p))
            if new_rows:
                rows = np.concatenate([rows, np.stack([eng.deim_row(r_) for r_ in new_rows])], axis=0)
                Pt.extend(new_rows)
            e_jump.append(jump)
            e_range.append(int(np.sum(e_jump)))
        self.geom_Pt = np.array(Pt)
        self.geom_alpha = np.array(e_points)
        self.geom_alpha_ranges = np.array(e_range)
        self.geom_interpol_verts = np.array(self.geom_interpol_verts)
        print("Computing interpolation elements utilizing differential operator, used", self.geom_alpha.shape[0],
              "constrained elements")

    def geom_constructed(self, r, case, interpol="geom"):
        """constraintsComponents.py:489-521: reconstruction of the train / test frames from the r leading basis blocks and
        the interpolation points: per dimension the normal equations of V_r[Pt] (r p x r p, host LU as the reference), the
        (e p x r p) by (r p x F) product on the device."""
        from scipy.linalg import lu_factor, lu_solve
        ns = self.nonlinearSnapshots
        eng, comm = ns._engine, ns._comm
        if comm.multi:
            raise NotImplementedError("geom_constructed runs on one rank")
        kind = getattr(self.param, "constProj_bases_interpolation_type", "")
        p = int(ns.constraintsSize) if kind in ("geom", "deim_block_form") else 1
        if case == "train":
            frames = ns.snapTensor
        elif case == "test":
            frames = ns.test_snapTensor
        else:
            raise ValueError("unknown frames to reconstruct.")
        F = frames.shape[0]
        if getattr(self.param, "constProj_snapshots_type", "") == "verts_bending":
            Pt = self.geom_Pt[:self.geom_alpha_ranges[r - 1]]
        else:
            Pt = self.geom_alpha[:self.geom_alpha_ranges[r - 1]]
        Pt = np.asarray(Pt, dtype=np.int64)
        rp = r * p
        if not self._comps_on_device:
            v0, n_loc = ns._shards[comm.rank]
            eng.components_upload(np.ascontiguousarray(self._comps[:, v0:v0 + n_loc, :]))
            self._comps_on_device = True
        VPt = np.stack([eng.deim_row(int(g)) for g in Pt])[:, :rp, :]            # (|Pt|, rp, 3)
        coef = np.empty((3, rp, F))
        for l in range(3):
            u, piv = lu_factor(VPt[:, :, l].T @ VPt[:, :, l])
            coef[l] = lu_solve((u, piv), VPt[:, :, l].T @ np.ascontiguousarray(frames[:, Pt, l]).T)
        return eng.components_expand(coef)

    @log_time(constProj_output_directory)
    def store_components_gradually_to_files(self, start, end, step, fileType):
        """constraintsComponents.py:572-594; fileType '.bin' or '.npy'."""
        from .utils import store_components, store_interpol_points_vector
        print('Storing bases ...', end='', flush=True)
        ns = self.nonlinearSnapshots
        numframes = ns.frs
        numverts = ns.num_constained_elements * ns.constraintsSize
        out = getattr(self.param, "constProj_output_directory", "")
        basesFile = os.path.join(out, self.fileNameBases)
        pointsFile = os.path.join(out, self.fileName_geom_points)
        vertsFile = os.path.join(out, "corrVerts")
        p = ns.constraintsSize
        for k in range(start, end + 1, step):
            store_components(basesFile, numframes, k * p, numverts, 3, self.comps[:k * p, :, :], fileType, 'Kp')
            store_interpol_points_vector(pointsFile, ns.frs, k, self.geom_alpha[:self.geom_alpha_ranges[k - 1]], fileType)
            store_interpol_points_vector(vertsFile, ns.frs, k, self.geom_interpol_verts[:k], fileType)
        print('done.')

    @log_time(constProj_output_directory)
    def store_components_n_interpol_points(self):
        """constraintsComponents.py:596-613: the .npz the PD simulator loads (Simulators.py:179-188)."""
        print('Storing bases and interpolation points to one file ...', end='', flush=True)
        data = {"components": self.comps, "interpol_alphas": self.geom_alpha, "Pt": self.geom_Pt,
                "interpol_verts": self.geom_interpol_verts, "interpol_alpha_ranges": self.geom_alpha_ranges}
        np.savez(os.path.join(self.param.constProj_output_directory,
                              "components_interpol_alphas_interpol_verts_interpol_alpha_ranges.npz"), **data)
        print("done!")
