"""``posComponents`` -- drop-in mirror of the reference class (snapbases/posComponents.py)
for the snapshot-reduction hot path, computing on the MI355X through ``libasb_hip.so``.

Same constructor (``posComponents(param)``), attribute and method names as the reference
(:19-49 and the methods listed in SURVEY.md 8b).  ``param`` is any object exposing the
attributes of the reference's ``Config_parameters`` that this path reads.

There is no CPU path: without the HIP library / a gfx950 GPU construction fails.
"""
import csv
import os

import numpy as np
from numpy import maximum, clip, sqrt, errstate, newaxis, empty, eye, allclose, dot, zeros, tensordot
from scipy.linalg import norm, svd, orth

from . import _lib
from .distributed import Comm
from .posSnapshots import posSnapshots
from . import utils as _u
from ._panels import deflate_panels_multirank
from .utils import log_time, store_components, testSparsity, test_linear_dependency


SMALL_TENSOR_BYTES = 256 << 20


def n_loc_of(snaps):
    return snaps._shards[snaps._comm.rank][1]


def _auto_global_mode(F, n_loc, multi):
    """Which device algorithm for support='global'.  The panel (projection) algorithm wins as soon as a read of X costs
    more than its per-panel overheads; a shard that fits the 256 MB Infinity Cache together with its residual copy
    (bunny: 68 MB) is faster through the plain residual loop, which has no host round trip at all (measured at
    14 290 x 200, K = 32: 1.3 ms against 2.8 ms).  Several ranks always use the panel protocol."""
    Fp = (int(F) + 15) // 16 * 16
    return "residual" if (not multi and 24 * int(n_loc) * Fp <= SMALL_TENSOR_BYTES) else "project"


class posComponents:  # Components == bases
    def __init__(self, param, pos_snapshots=None):
        self.basesType = param.vertPos_bases_type
        assert self.basesType == 'PCA' or self.basesType == 'SPLOCS'

        if pos_snapshots is None:
            train = os.path.join(param.aligned_snapshots_directory, param.train_aligned_snapshots_animation_file)
            test = os.path.join(param.aligned_snapshots_directory, param.test_aligned_snapshots_animation_file)
            pos_snapshots = posSnapshots(train, test, param.vertPos_rest_shape, param.vertPos_masses_file,
                                         param.tet_mesh_file, param.q_standarize, param.q_massWeight)
        self.pos_snapshots = pos_snapshots

        self.numComp = param.vertPos_numComponents
        self.support = param.q_support              # 'local' | 'global'
        self.storeSingVal = param.store_vertPos_PCA_sing_val

        self._comps = None            # host copy (cache of the device-resident basis, or user-assigned)
        self._comps_on_device = False  # True: the authoritative (K, n_loc, 3) basis is in HBM
        self.weigs = None
        self.ortho_comps = None
        self.smooth_min_dist = param.vertPos_smooth_min_dist
        self.smooth_max_dist = param.vertPos_smooth_max_dist
        self.output_components_file = "components.h5"

        self.measures_at_largeDeforVerts = None
        self.fileNameBases = "q_pos_"
        self.param = param

        # extras (not in the reference): the selected vertices; which device algorithm to use
        # (None = automatic, "residual", "project"; env ASB_DEFLATE_MODE overrides the default)
        self.selected_vertices = None
        self.deflate_mode = os.environ.get("ASB_DEFLATE_MODE") or None
        self._stepwise_panels = False      # tests: drive the multi-rank panel protocol on one rank

    # ------------------------------------------------------------------ comps: device-resident, lazy download
    @property
    def comps(self):
        if self._comps is None and self._comps_on_device:
            eng, comm = self.pos_snapshots._engine, self.pos_snapshots._comm
            if getattr(self, "_comps_streamed", False) and getattr(eng, "_streaming", False) and not comm.multi:
                # the engine streamed the rows into its pinned buffer while it computed (HipEngine.components_stream): a plain
                # ndarray over that buffer, valid until the next run on this engine (np.array(...) to keep it longer)
                self._comps = eng.components_pinned()
                return self._comps
            loc = eng.results(want_comps=True, want_weigs=False)["comps"]
            self._comps = comm.all_gather_rows(loc, self.pos_snapshots.nVerts, axis=1)
        return self._comps

    @comps.setter
    def comps(self, value):
        self._comps = value
        self._comps_on_device = False

    # ------------------------------------------------------------------ statics (same as the reference)
    @staticmethod
    def project_weight(x):
        """posComponents.py:52-58."""
        x = maximum(0., x)
        max_x = x.max()
        if max_x == 0:
            return x
        else:
            return x / max_x

    @staticmethod
    def compute_support_map(idx, geodesics, min_dist, max_dist):
        """posComponents.py:61-64."""
        phi = geodesics(idx)
        return (clip(phi, min_dist, max_dist) - min_dist) / (max_dist - min_dist)

    @staticmethod
    def prox_l1l2(Lambda, x, beta):
        """posComponents.py:252-256."""
        xlen = sqrt((x ** 2).sum(axis=-1))
        with errstate(divide='ignore'):
            shrinkage = maximum(0.0, 1 - beta * Lambda / xlen)
        return x * shrinkage[..., newaxis]

    # ------------------------------------------------------------------ the hot path
    @log_time("")
    def extract_k_components(self, writer, num_iters_max=20, num_admm_iterations=10):
        """posComponents.py:67-129 on the GPU.

        support='global' on one rank runs all K components back to back on the device;
        support='local' (host geodesic support map between select and apply) and multi-rank
        runs step through select -> [all-gather] -> pick -> apply per component.
        """
        snaps = self.pos_snapshots
        eng, comm = snaps._engine, snaps._comm
        K = self.numComp
        local = self.support == 'local'
        if local and snaps.compute_geodesic_distance is None:
            raise ValueError("support='local' needs the mesh triangles (geodesic support maps)")
        # global support on one rank: residual-free panel algorithm (one read of X per panel of up to
        # 16 components); otherwise the residual tensor is kept and updated per component.
        mode = self.deflate_mode
        if mode is None:
            mode = "residual" if local else _auto_global_mode(self.pos_snapshots.frs, n_loc_of(snaps), comm.multi)
        if mode == "project" and local:
            raise ValueError("deflate_mode='project' needs global support")
        eng.deflate_begin(K, local, _lib.DEFLATE_PROJECT if mode == "project" else _lib.DEFLATE_RESIDUAL)
        v0, n_loc = snaps._shards[comm.rank]

        if not comm.multi and not local and not self._stepwise_panels:
            eng.run_global(0, K)
        elif mode == "project":
            self._project_multirank(K)
        else:
            rec = recs = None
            if comm.multi:
                rec, recs = comm.new_records(eng.xchg_len(), eng.device_exchange)
            geo = snaps.compute_geodesic_distance
            if local and hasattr(geo, "prepare"):
                geo.prepare()                         # (lazy: built on first use, posSnapshots only records mesh and backend)
            on_device_maps = local and getattr(geo, "_engine", None) is eng and getattr(eng, "geodesic_dense", False) \
                and type(self).compute_support_map is posComponents.compute_support_map
            for k in range(K):
                if comm.multi:
                    eng.local_best(k, rec.data_ptr())
                    comm.all_gather_records(rec, recs)
                    eng.pick(k, recs.data_ptr(), comm.world)
                else:
                    eng.pick(k)
                s_loc = None
                if local and on_device_maps:          # pick -> distance field -> support weights -> pass: no host round trip
                    eng.apply_geodesic(k, self.smooth_min_dist, self.smooth_max_dist)
                    continue
                if local:
                    idx, _ = eng.get_pick(k)
                    s = 1 - self.compute_support_map(idx, snaps.compute_geodesic_distance,
                                                     self.smooth_min_dist, self.smooth_max_dist)      # (N,)
                    s_loc = s[v0:v0 + n_loc]
                eng.apply(k, s_loc)

        res = eng.results(want_comps=False, want_weigs=True)
        normR = np.sqrt(np.maximum(comm.allreduce_sum(res["normR2_local"]), 0.0))     # |X|^2 - sum can round below 0 once |R| < 1e-8 |X|
        self.weigs = res["weigs"]
        self.selected_vertices = res["idx"]
        self._comps, self._comps_on_device = None, True
        self._comps_streamed = True     # the engine's pinned buffer (if it streams) holds exactly this run's basis
        self.measures_at_largeDeforVerts = np.column_stack([np.arange(K, dtype=np.float64), res["sigma"], normR])
        if self.storeSingVal and writer is not None:
            for k in range(K):
                writer.writerow([k, float(res["sigma"][k]), float(normR[k])])

        if self.basesType == 'SPLOCS':
            self.splocs_glob_optimization(self.param.splocs_max_itrs, self.param.splocs_admm_num_itrs,
                                          None, snaps.compute_geodesic_distance)
        print("Computed '", self.basesType, "' bases size ", (K, snaps.nVerts, 3))

    def _project_multirank(self, K):
        snaps = self.pos_snapshots
        deflate_panels_multirank(snaps._engine, snaps._comm, snaps.nVerts, K)

    @log_time("")
    def splocs_glob_optimization(self, num_iters_max, num_admm_iterations, R, compute_geodesic_distance):
        with _u.no_gc():          # (a garbage collection inside the loop is 2 - 3 ms of idle GPU: utils.no_gc)
            return self._splocs_glob_optimization(num_iters_max, num_admm_iterations, R, compute_geodesic_distance)

    def _splocs_glob_optimization(self, num_iters_max, num_admm_iterations, R, compute_geodesic_distance):
        """posComponents.py:131-189 on the GPU (csrc/asb_splocs.hip).

        Like the reference, this leaves ``comps`` / ``weigs`` untouched (the reference works on
        local copies and discards them) and prints one
        ``itr %03d, Energy =%f, Error =%f`` line per outer iteration.  Extras: the refined
        ``splocs_comps`` (K,N,3), ``splocs_weigs`` (F,K), ``splocs_trace`` (its,2) and
        ``splocs_centres`` (its,K) attributes.  ``R`` is accepted for signature compatibility;
        the device works from X, W and C (the residual is never formed).
        """
        snaps = self.pos_snapshots
        eng, comm = snaps._engine, snaps._comm
        K, N, F = self.numComp, snaps.nVerts, snaps.frs
        v0, n_loc = snaps._shards[comm.rank]
        lam, rho = self.param.splocs_lambda, self.param.splocs_rho
        if compute_geodesic_distance is None:
            raise ValueError("SPLOCS needs the mesh triangles (geodesic support maps)")
        eng.splocs_begin()
        Pbuf = Mbuf = None
        if comm.multi:
            Pbuf, Mbuf = comm.new_gram_buffers(F, K, eng.device_exchange)

        def gram(want_norm=False):
            nx = eng.splocs_gram(Pbuf.data_ptr() if Pbuf is not None else None,
                                 Mbuf.data_ptr() if Mbuf is not None else None, want_norm)
            if comm.multi:                 # partial Gram matrices: RCCL all-reduce (sum) over the ranks
                comm.allreduce_tensor(Pbuf)
                comm.allreduce_tensor(Mbuf)
            return nx

        def ptrs():
            return (Pbuf.data_ptr(), Mbuf.data_ptr()) if Pbuf is not None else (None, None)

        normX2 = comm.allreduce_sum(gram(True))[0]
        phi_cache = {}
        # distance fields and support maps stay on the device when the geodesics run on this engine
        fields_on_device = getattr(compute_geodesic_distance, "_engine", None) is eng and hasattr(eng, "splocs_admm_fields")
        slot_of = {}
        if fields_on_device:
            eng.geodesic_cache_clear()
        trace, centres = [], []
        # the objective of every iteration stays on the device and is read once after the loop (nothing in the loop depends
        # on the printed numbers): with the distance fields on the device an outer iteration synchronises once, for its centres
        deferred = fields_on_device and hasattr(eng, "splocs_trace_begin") and num_iters_max > 0 and \
            os.environ.get("ASB_SPLOCS_DEFER", "1") != "0"
        if deferred:
            eng.splocs_trace_begin(num_iters_max)
        for it in range(num_iters_max):
            cidx, cval = eng.splocs_weights(*ptrs())                          # :144-156, :161
            if comm.multi:
                cidx = comm.global_argmax(cidx, cval)
            if fields_on_device:
                wanted = [int(i) for i in dict.fromkeys(cidx.tolist())]
                missing = [i for i in wanted if i not in slot_of]
                if len(slot_of) + len(missing) > eng.GEODESIC_CACHE_SLOTS:    # cache full: start again with this iteration's centres
                    eng.geodesic_cache_clear()
                    slot_of, missing = {}, wanted
                if missing:
                    tol = getattr(compute_geodesic_distance, "_tol", 1e-13)
                    slot_of.update(zip(missing, eng.geodesic_cache_add(missing, tol)))
                eng.splocs_admm_fields([slot_of[int(i)] for i in cidx], lam, self.smooth_min_dist, self.smooth_max_dist,
                                       rho, num_admm_iterations)              # :162-181
            else:
                missing = [int(i) for i in dict.fromkeys(cidx.tolist()) if int(i) not in phi_cache]
                if missing:
                    many = getattr(compute_geodesic_distance, "solve_many", None)
                    phis = many(missing) if many else [compute_geodesic_distance(i) for i in missing]
                    for i, phi in zip(missing, phis):
                        phi_cache[i] = phi
                Lambda = np.empty((K, n_loc))
                for k in range(K):                                            # :162-165
                    phi = phi_cache[int(cidx[k])]
                    smap = (clip(phi, self.smooth_min_dist, self.smooth_max_dist) - self.smooth_min_dist) \
                        / (self.smooth_max_dist - self.smooth_min_dist)
                    Lambda[k] = lam * smap[v0:v0 + n_loc]
                eng.splocs_admm(Lambda, rho, num_admm_iterations)             # :167-181
            gram()
            centres.append(cidx.copy())
            if deferred:
                eng.splocs_objective_dev(it, *ptrs())
                continue
            wp, gm, sp = eng.splocs_objective(*ptrs())
            sparsity = comm.allreduce_sum(sp)[0]
            r2 = max(normX2 - 2.0 * wp + gm, 0.0)                             # |X - W C|^2
            E_rms = np.sqrt(r2) / sqrt(3 * N * F)
            energy = r2 + sparsity
            trace.append([energy, E_rms])
            if comm.rank == 0:
                print("itr %03d, Energy =%f, Error =%f" % (it, energy, E_rms))
        if deferred:
            tr = eng.splocs_trace(num_iters_max)                              # (its, 3): <W,P>, <G,M>, local sum Lambda |C_v|
            spars = comm.allreduce_sum(tr[:, 2].copy())
            for it in range(num_iters_max):
                r2 = max(normX2 - 2.0 * tr[it, 0] + tr[it, 1], 0.0)
                E_rms = np.sqrt(r2) / sqrt(3 * N * F)
                energy = r2 + spars[it]
                trace.append([energy, E_rms])
                if comm.rank == 0:
                    print("itr %03d, Energy =%f, Error =%f" % (it, energy, E_rms))
        C_loc, W_new = eng.splocs_results()
        self.splocs_comps = comm.all_gather_rows(C_loc, N, axis=1)
        self.splocs_weigs = W_new
        self.splocs_trace = np.array(trace)
        self.splocs_centres = np.array(centres, dtype=np.int64)

    @log_time("")
    def compute_components_store_singvalues(self):
        """posComponents.py:258-272."""
        headerSing = ['component', 'singVal', 'norm_R']
        file_name = os.path.join(self.param.vertPos_output_directory,
                                 self.param.name + "_posBases_pcaExtraction_singValues_errorNorm")
        if self.storeSingVal:
            if self.pos_snapshots._comm.rank == 0:
                with open(file_name + '.csv', 'w', encoding='UTF8') as singFile:
                    writer = csv.writer(singFile)
                    writer.writerow(headerSing)
                    self.extract_k_components(writer)
            else:
                self.extract_k_components(None)
        else:
            self.extract_k_components(None)

    @log_time("")
    def post_process_components(self):
        """posComponents.py:274-302 on the device-resident basis: un-scale and add the mean
        (:279-282), per-dimension `orth` (:284-287: Gram -> K x K eigen-solve -> U = A V S^-1, the
        partial Gram matrices all-reduced over ranks), mass un-weighting (:289-292), then the
        reference's printed checks."""
        print("Post-processing pos components ...")
        snaps = self.pos_snapshots
        eng, comm = snaps._engine, snaps._comm
        v0, n_loc = snaps._shards[comm.rank]
        self._comps_streamed = False    # (the device basis is about to change: a later read copies it afresh)
        if not self._comps_on_device:
            if self._comps is None:
                raise ValueError("no components: run compute_components_store_singvalues first")
            eng.components_upload(np.ascontiguousarray(self._comps[:, v0:v0 + n_loc, :]))      # caller-assigned comps
            self._comps_on_device = True
        later = bool(self.param.q_orthogonal or self.param.q_massWeight)       # (the basis is downloaded once, behind the last step)
        loc = eng.components_post(self.param.q_standarize, snaps.pre_scale_factor, None, download=not later)
        if self.param.q_orthogonal:
            K = self.numComp
            # Gram -> K x K Jacobi eigen-solver -> U = A V S^-1, all on the device (any K)
            Gbuf = comm.new_buffer(3 * K * K, eng.device_exchange) if comm.multi else None
            gp = Gbuf.data_ptr() if Gbuf is not None else None
            eng.orth_gram(gp)
            if Gbuf is not None:
                comm.allreduce_tensor(Gbuf)
            self.ortho_sing_vals = eng.orth_apply(gp)
            # the Gram route leaves U^T U = I + O(eps cond^2): one Newton-Schulz step with the Gram of U removes it
            Gbuf = comm.new_buffer(3 * K * K, eng.device_exchange) if comm.multi else None
            gp = Gbuf.data_ptr() if Gbuf is not None else None
            eng.orth_gram(gp)
            if Gbuf is not None:
                comm.allreduce_tensor(Gbuf)
            eng.orth_refine(gp)
            loc = None
        if self.param.q_massWeight:
            assert snaps.nVerts == snaps.invMassL.shape[0]
            loc = eng.components_post(False, 1.0, snaps.invMassL[v0:v0 + n_loc])
        if loc is None:
            loc = eng.results(want_comps=True, want_weigs=False)["comps"]
        self._comps = comm.all_gather_rows(loc, snaps.nVerts, axis=1)

        testSparsity(self.comps)
        test_linear_dependency(self.comps, 3, self.numComp)
        if self.param.q_orthogonal:
            self.is_utmu_orthogonal()
        print("... Volkwein (" + str(self.param.q_massWeight) + ")... standerized (" + str(self.param.q_standarize) +
              ")... support (" + str(self.support) + "), orthogonalized (" + str(self.param.q_orthogonal) + ").")

    @log_time("")
    def is_utmu_orthogonal(self):
        """posComponents.py:304-313."""
        print('... testing M orthogonality, U^T M U = I (K x K) ...', end='', flush=True)
        mass = self.pos_snapshots.mass
        if mass is None:        # (the reference needs q_massWeight here; without masses check U^T U = I)
            mass = np.ones(self.comps.shape[1])
        for l in range(self.comps.shape[2]):
            Mu_l = self.comps[:, :, l].T * mass[:, None]
            utMu_l = dot(self.comps[:, :, l], Mu_l)
            assert allclose(utMu_l, eye(self.comps.shape[0]))
        print('(True).')

    @log_time("")
    def store_components_to_files(self, start, end, step, fileType):
        """posComponents.py:315-327; fileType '.bin' or '.npy'."""
        print('Storing bases ...', end='', flush=True)
        numframes, numverts = self.pos_snapshots.frs, self.pos_snapshots.nVerts
        basesFile = os.path.join(self.param.vertPos_output_directory, self.fileNameBases)
        for k in range(start, end + 1, step):
            store_components(basesFile, numframes, k, numverts, 3, self.comps[:k, :, :], fileType, 'K')
        print('done.')

    @log_time("")
    def store_animations(self, output_bases_dir):
        """posComponents.py:329-341 (needs h5py, like the reference)."""
        import h5py
        output_components = os.path.join(output_bases_dir, self.output_components_file)
        with h5py.File(output_components, 'w') as f:
            f['default'] = self.pos_snapshots.verts[0]
            f['tris'] = self.pos_snapshots.tris
            for i, c in enumerate(self.comps):
                f['comp%03d' % i] = c

    @log_time("")
    def test_basesSingVals(self):
        """posComponents.py:343-356."""
        bases = self.comps.copy()
        s = empty((bases.shape[0], 3))
        for i in range(3):
            sing = svd(bases[:, :, i], full_matrices=False, compute_uv=False)
            s[:, i] = sing / sing.max()
        return s

    @log_time("")
    def test_convergence(self, start, end, step, writer=None):
        """posComponents.py:191-214."""
        snapshots = self.pos_snapshots.snapTensor.copy()
        fro_err, rel_err_x, rel_err_y, rel_err_z, max_err = [], [], [], [], []
        for k in range(start, end + 1, step):
            reconstructed = tensordot(self.weigs[:, :k], self.comps[:k, :, :], axes=([1], [0]))
            fro_err.append(self.frobenius_error(snapshots, reconstructed))
            rel_err = self.relative_error_per_component(snapshots, reconstructed)
            rel_err_x.append(rel_err[0])
            rel_err_y.append(rel_err[1])
            rel_err_z.append(rel_err[2])
            max_err.append(self.max_pointwise_error(snapshots, reconstructed))
        return fro_err, max_err, rel_err_x, rel_err_y, rel_err_z

    @staticmethod
    def frobenius_error(f, f_reconstructed):
        """posComponents.py:217-223."""
        return norm(f - f_reconstructed)

    @staticmethod
    def relative_error_per_component(f, f_reconstructed):
        """posComponents.py:225-237."""
        return [norm(f[:, :, i] - f_reconstructed[:, :, i]) / norm(f[:, :, i]) for i in range(3)]

    @staticmethod
    def max_pointwise_error(f, f_reconstructed):
        """posComponents.py:239-249."""
        return np.max(np.abs(f - f_reconstructed)) / np.max(f)
