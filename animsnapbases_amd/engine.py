"""One-shard device engine: an object wrapper over the C ABI of ``libasb_hip.so``.

``HipEngine`` is the ONLY compute engine of the product.  The host classes
(``posSnapshots`` / ``posComponents``) talk to it through the small method set below, so
the multi-rank host logic can be exercised on CPU by a test double that lives in
``tests/`` (never in this package).
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from ._lib import AsbLibraryError, ptr


class _PinnedBasis(object):
    """Pinned host memory for a streamed basis (asb_host_alloc).  The ndarray views ``HipEngine.components_pinned`` hands out
    keep this object alive through their buffer, so the memory is freed when the engine AND every view have let go of it --
    never under a live view (the context itself never frees a caller-owned buffer: asb_components_stream_into)."""

    def __init__(self, lib, count):
        p = ctypes.c_void_p()
        if lib.asb_host_alloc(int(count), ctypes.byref(p)) != 0 or not p.value:
            raise MemoryError("asb_host_alloc(%d doubles) failed" % count)
        self._lib, self.ptr, self.count = lib, p.value, int(count)
        self._views = []            # weak references to the ctypes arrays the handed-out ndarrays are built on

    def view(self, shape):
        n = int(np.prod(shape))
        assert n <= self.count
        buf = (ctypes.c_double * n).from_address(self.ptr)
        buf._owner = self           # buffer -> owner: the owner lives as long as any ndarray over it
        self._views = [w for w in self._views if w() is not None] + [weakref.ref(buf)]
        return np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape)

    def in_use(self):
        self._views = [w for w in self._views if w() is not None]
        return bool(self._views)

    def __del__(self):
        try:
            if self.ptr:
                self._lib.asb_host_free(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


class HipEngine(object):
    """Owns one ``asb_ctx`` = one GPU, one HIP stream, one vertex shard."""

    device_exchange = True      # exchange records live in device memory (torch tensors)

    STREAM_DEFAULT = -1         # ASB_STREAM_DEFAULT: the device's null stream

    def __init__(self, device_id=0, stream=None):
        """stream: None = private stream; an int hipStream_t handle; 0 or STREAM_DEFAULT = the null
        stream (torch's current stream when the caller has not switched streams)."""
        self.lib = _lib.load()
        if self.lib.asb_abi_version() != 1:
            raise AsbLibraryError("libasb_hip.so ABI version mismatch")
        h = ctypes.c_void_p()
        if stream is None:
            sarg = None
        elif stream in (0, self.STREAM_DEFAULT):
            sarg = ctypes.c_void_p(-1)
        else:
            sarg = ctypes.c_void_p(stream)
        rc = self.lib.asb_create(int(device_id), sarg, ctypes.byref(h))
        if rc != 0:
            msg = self.lib.asb_last_error(h).decode() if h else ""
            if h:
                self.lib.asb_destroy(h)
            raise AsbLibraryError("asb_create(device %d) failed with status %d %s -- a gfx950 (MI355X) GPU is "
                                  "required; there is no CPU fallback" % (device_id, rc, msg))
        self.h = h
        self.device_id = int(device_id)
        # the stream the context enqueues on, as torch would name it: None = private, 0 = the null stream, else the handle
        self.stream_handle = None if stream is None else (0 if stream in (0, self.STREAM_DEFAULT) else int(stream))
        self.F = self.n_loc = self.v0 = self.N_glob = self.K = 0

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("libasb_hip: status %d: %s" % (rc, self.lib.asb_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.asb_destroy(self.h)      # (synchronises the copy stream; a caller-owned pinned buffer is not freed there)
            self.h = None
        self._pin = None                      # the views that are still alive keep their memory

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._ck(self.lib.asb_sync(self.h))

    def prof_reset(self, enable=True):
        self._ck(self.lib.asb_prof_reset(self.h, int(bool(enable))))

    def prof_get(self):
        n, ms = ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.asb_prof_get(self.h, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    # ------------------------------------------------------------------ snapshots
    def upload(self, X, v0, n_loc, massL=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        F, N, three = X.shape
        assert three == 3
        if massL is not None:
            massL = np.ascontiguousarray(massL, dtype=np.float64)
            assert massL.shape == (N,)
        self._ck(self.lib.asb_snapshots_upload(self.h, ptr(X), F, N, int(v0), int(n_loc), ptr(massL)))
        self.F, self.N_glob, self.v0, self.n_loc = F, N, int(v0), int(n_loc)

    def adopt_device(self, dev_ptr, F, n_loc, massL_loc=None, v0=0, N_glob=None):
        """dev_ptr: device address of an (F, n_loc, 3) float64 tensor (reference layout) holding this
        rank's vertices [v0, v0 + n_loc) of N_glob."""
        if massL_loc is not None:
            massL_loc = np.ascontiguousarray(massL_loc, dtype=np.float64)
        N_glob = int(n_loc if N_glob is None else N_glob)
        self._ck(self.lib.asb_snapshots_adopt_dev(self.h, ctypes.c_void_p(dev_ptr), int(F), int(n_loc), ptr(massL_loc),
                                                  int(v0), N_glob))
        self.F, self.N_glob, self.v0, self.n_loc = int(F), N_glob, int(v0), int(n_loc)

    def upload_rest(self, X, v0, n_loc, massL, rest_shape_code, subtract):
        """upload + rest shape in one sweep: (sum(x), sum(x^2) or None)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        F, N, three = X.shape
        assert three == 3
        if massL is not None:
            massL = np.ascontiguousarray(massL, dtype=np.float64)
            assert massL.shape == (N,)
        sums = np.zeros(2)
        self._ck(self.lib.asb_snapshots_upload_rest(self.h, ptr(X), F, N, int(v0), int(n_loc), ptr(massL), int(rest_shape_code),
                                                    int(bool(subtract)), ptr(sums)))
        self.F, self.N_glob, self.v0, self.n_loc = F, N, int(v0), int(n_loc)
        return float(sums[0]), (float(sums[1]) if sums[1] >= 0 else None)

    def adopt_device_rest(self, dev_ptr, F, n_loc, massL_loc, v0, N_glob, rest_shape_code, subtract):
        if massL_loc is not None:
            massL_loc = np.ascontiguousarray(massL_loc, dtype=np.float64)
        N_glob = int(n_loc if N_glob is None else N_glob)
        sums = np.zeros(2)
        self._ck(self.lib.asb_snapshots_adopt_dev_rest(self.h, ctypes.c_void_p(dev_ptr), int(F), int(n_loc), ptr(massL_loc),
                                                       int(v0), N_glob, int(rest_shape_code), int(bool(subtract)), ptr(sums)))
        self.F, self.N_glob, self.v0, self.n_loc = int(F), N_glob, int(v0), int(n_loc)
        return float(sums[0]), (float(sums[1]) if sums[1] >= 0 else None)

    def center(self, rest_shape_code, subtract):
        s = ctypes.c_double()
        self._ck(self.lib.asb_snapshots_center(self.h, int(rest_shape_code), int(bool(subtract)), ctypes.byref(s)))
        return s.value

    def sqdev(self, mu):
        s = ctypes.c_double()
        self._ck(self.lib.asb_snapshots_sqdev(self.h, float(mu), ctypes.byref(s)))
        return s.value

    def scale(self, a):
        self._ck(self.lib.asb_snapshots_scale(self.h, float(a)))

    def get_mean(self):
        out = np.empty((self.n_loc, 3))
        self._ck(self.lib.asb_snapshots_get_mean(self.h, ptr(out)))
        return out

    def download_snapshots(self):
        out = np.empty((self.F, self.n_loc, 3))
        self._ck(self.lib.asb_snapshots_download(self.h, ptr(out)))
        return out

    # ------------------------------------------------------------------ deflation
    def deflate_begin(self, K, local_support, mode=_lib.DEFLATE_RESIDUAL):
        if getattr(self, "_streaming", False):
            # the streamed basis goes into pinned memory THIS side owns: a buffer some ndarray still looks at is neither
            # freed nor overwritten by the new run -- the run gets a fresh one and the old one dies with its last view
            need = int(K) * int(self.n_loc) * 3
            pin = getattr(self, "_pin", None)
            if need > 0 and (pin is None or pin.count < need or pin.in_use()):
                self._pin = pin = _PinnedBasis(self.lib, need)
            if need > 0:
                self._ck(self.lib.asb_components_stream_into(self.h, ctypes.c_void_p(pin.ptr), pin.count))
        self._ck(self.lib.asb_deflate_begin(self.h, int(K), int(mode), int(bool(local_support))))
        self.K = int(K)
        self.mode = int(mode)

    def xchg_len(self):
        return int(self.lib.asb_deflate_xchg_len(self.h))

    def local_best(self, k, rec_dev_ptr):
        self._ck(self.lib.asb_deflate_local_best(self.h, int(k), ctypes.c_void_p(rec_dev_ptr)))

    def pick(self, k, recs_dev_ptr=None, n_rec=0):
        self._ck(self.lib.asb_deflate_pick(self.h, int(k), ctypes.c_void_p(recs_dev_ptr) if recs_dev_ptr else None,
                                           int(n_rec)))

    def get_pick(self, k):
        i, s = ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.asb_deflate_get_pick(self.h, int(k), ctypes.byref(i), ctypes.byref(s)))
        return i.value, s.value

    def block_argmax(self, p):
        """(global block index, energy) of the constraint block with the largest residual energy on this shard."""
        i, v = ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.asb_deflate_block_argmax(self.h, int(p), ctypes.byref(i), ctypes.byref(v)))
        return i.value, v.value

    def force_next(self, gidx):
        self._ck(self.lib.asb_deflate_force_next(self.h, int(gidx)))

    def apply(self, k, s_loc=None):
        if s_loc is not None:
            s_loc = np.ascontiguousarray(s_loc, dtype=np.float64)
            assert s_loc.shape == (self.n_loc,)
        self._ck(self.lib.asb_deflate_apply(self.h, int(k), ptr(s_loc)))

    def run_global(self, k0, k1):
        self._ck(self.lib.asb_deflate_run_global(self.h, int(k0), int(k1)))

    def results(self, want_comps=True, want_weigs=True):
        K = self.K
        comps = np.empty((K, self.n_loc, 3)) if want_comps else None
        weigs = np.empty((self.F, K)) if want_weigs else None
        idx = np.empty(K, dtype=np.int64)
        sigma = np.empty(K)
        nr2 = np.empty(K)
        self._ck(self.lib.asb_deflate_results(self.h, ptr(comps), ptr(weigs), ptr(idx), ptr(sigma), ptr(nr2)))
        return dict(comps=comps, weigs=weigs, idx=idx, sigma=sigma, normR2_local=nr2)

    # ------------------------------------------------------------------ projection mode, panel steps
    NBINS = 2048

    def panel_scale(self, set_e0max=-1.0):
        a, b = ctypes.c_double(), ctypes.c_double()
        self._ck(self.lib.asb_panel_scale(self.h, ctypes.byref(a), ctypes.byref(b), float(set_e0max)))
        return a.value, b.value

    def panel_guess_stats(self):
        a, b, ok = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._ck(self.lib.asb_panel_guess_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(ok)))
        return a.value, b.value, bool(ok.value)

    def panel_guess_begin(self, world):
        self._ck(self.lib.asb_panel_guess_begin(self.h, int(world)))

    def panel_guess_end(self):
        self._ck(self.lib.asb_panel_guess_end(self.h))

    def panel_sub_run(self, sp, k0, steps, spec_max):
        ran, proven, cont = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        self._ck(self.lib.asb_panel_sub_run(self.h, int(sp), int(k0), int(steps), int(spec_max), ctypes.byref(ran), ctypes.byref(proven),
                                            ctypes.byref(cont)))
        return ran.value, proven.value, bool(cont.value)

    def panel_sub_project(self, k0, ncs):
        arr = (ctypes.c_int * len(ncs))(*[int(x) for x in ncs])
        self._ck(self.lib.asb_panel_sub_project(self.h, int(k0), len(ncs), arr))

    def panel_sub_check(self, ct, kb, nc, out_dev_ptr):
        self._ck(self.lib.asb_panel_sub_check(self.h, int(ct), int(kb), int(nc), ctypes.c_void_p(out_dev_ptr)))

    def panel_sub_commit(self, ct, kb, nc, kept):
        self._ck(self.lib.asb_panel_sub_commit(self.h, int(ct), int(kb), int(nc), int(kept)))

    def panel_hist(self, level, hist_ptr=None):
        self._ck(self.lib.asb_panel_hist(self.h, int(level), ctypes.c_void_p(hist_ptr) if hist_ptr else None))

    def panel_tau(self, level, hist_ptr=None):
        self._ck(self.lib.asb_panel_tau(self.h, int(level), ctypes.c_void_p(hist_ptr) if hist_ptr else None))

    def panel_top_energies(self, out_ptr, cap):
        self._ck(self.lib.asb_panel_top_energies(self.h, ctypes.c_void_p(out_ptr), int(cap)))

    def panel_global_tau(self, tab_ptr, world, cap):
        """Per-rank candidate counts after installing the global threshold; None when the table is too large for the
        device selection (the caller then selects with torch and calls panel_set_tau)."""
        counts = np.empty(world, dtype=np.int64)
        rc = self.lib.asb_panel_global_tau(self.h, ctypes.c_void_p(tab_ptr), int(world), int(cap), ptr(counts))
        if rc == _lib.ERR_LIMIT:
            return None
        self._ck(rc)
        return counts

    def panel_set_tau(self, tau_ptr):
        self._ck(self.lib.asb_panel_set_tau(self.h, ctypes.c_void_p(tau_ptr)))

    def panel_target(self):
        return int(self.lib.asb_panel_target(self.h))

    def panel_capacity(self):
        return int(self.lib.asb_panel_capacity(self.h))

    def panel_row_len(self):
        return 3 * ((self.F + 15) // 16 * 16)

    def panel_select(self, k, rows_ptr, idx_ptr, forced_gidx=-1, global_all=False, want_counts=True):
        if not want_counts:          # no host synchronisation: the caller knows the counts from the gathered energies
            self._ck(self.lib.asb_panel_select(self.h, int(k), int(forced_gidx), int(bool(global_all)),
                                               ctypes.c_void_p(rows_ptr), ctypes.c_void_p(idx_ptr), None, None))
            return None, None
        n, ov = ctypes.c_int64(), ctypes.c_int()
        self._ck(self.lib.asb_panel_select(self.h, int(k), int(forced_gidx), int(bool(global_all)),
                                           ctypes.c_void_p(rows_ptr), ctypes.c_void_p(idx_ptr), ctypes.byref(n),
                                           ctypes.byref(ov)))
        return n.value, bool(ov.value)

    def panel_assemble(self, rows_g_ptr, idx_g_ptr, counts, maxcount):
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        self._ck(self.lib.asb_panel_assemble(self.h, ctypes.c_void_p(rows_g_ptr), ctypes.c_void_p(idx_g_ptr), ptr(counts),
                                             int(counts.shape[0]), int(maxcount)))

    def panel_assemble_packed(self, packed_g_ptr, counts, maxcount):
        """One all-gathered buffer: per rank ``maxcount`` rows followed by ``maxcount`` vertex ids."""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        self._ck(self.lib.asb_panel_assemble_packed(self.h, ctypes.c_void_p(packed_g_ptr), ptr(counts), int(counts.shape[0]),
                                                    int(maxcount)))

    def panel_run(self, k0, steps, global_all=False):
        c = ctypes.c_int64()
        self._ck(self.lib.asb_panel_run(self.h, int(k0), int(steps), int(bool(global_all)), 1, ctypes.byref(c)))
        return c.value

    def panel_project(self, k0, ncols):
        self._ck(self.lib.asb_panel_project(self.h, int(k0), int(ncols)))

    def panel_run_spec(self, k0, steps, global_all, spec_max):
        """panel_run that may append up to ``spec_max`` unproven steps: (steps run, provable head)."""
        ran, proven = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.lib.asb_panel_run_spec(self.h, int(k0), int(steps), int(bool(global_all)), 1, int(spec_max),
                                             ctypes.byref(ran), ctypes.byref(proven)))
        return ran.value, proven.value

    def panel_project_spec(self, k0, ncols, proven):
        """Pass over X for all ``ncols`` steps, energies untouched; first unproven step this shard rejects (ncols: none)."""
        r = ctypes.c_int64()
        self._ck(self.lib.asb_panel_project_spec(self.h, int(k0), int(ncols), int(proven), ctypes.byref(r)))
        return r.value

    def panel_project_spec_dev(self, k0, ncols, proven, out_dev_ptr):
        self._ck(self.lib.asb_panel_project_spec_dev(self.h, int(k0), int(ncols), int(proven), ctypes.c_void_p(out_dev_ptr)))

    def fetch_double(self, dev_ptr):
        v = ctypes.c_double()
        self._ck(self.lib.asb_fetch_double(self.h, ctypes.c_void_p(dev_ptr), ctypes.byref(v)))
        return v.value

    def fetch_doubles(self, dev_ptr, n):
        out = np.empty(int(n))
        self._ck(self.lib.asb_fetch_doubles(self.h, ctypes.c_void_p(dev_ptr), int(n), ptr(out)))
        return out

    def panel_read_run(self, k0, k1, nsub_max, spec_budget, sub_budget, words_dev_ptr):
        """All sub-panels of a multi-rank read in one launch, pass and tile checks behind it (asb.h: asb_panel_read_run):
        (ntile, [columns per tile], [provable head per tile]); the verdict words are left at words_dev_ptr (10 doubles)."""
        nt = ctypes.c_int(0)
        nc, pr = (ctypes.c_int * 8)(), (ctypes.c_int * 8)()
        sb = (ctypes.c_int * 8)(*[int(x) for x in sub_budget])
        self._ck(self.lib.asb_panel_read_run(self.h, int(k0), int(k1), int(nsub_max), int(spec_budget), sb,
                                             ctypes.c_void_p(words_dev_ptr), ctypes.byref(nt), nc, pr))
        return nt.value, list(nc)[:nt.value], list(pr)[:nt.value]

    def panel_read_commit(self, words10):
        words10 = np.ascontiguousarray(words10, dtype=np.float64)
        assert words10.shape == (10,)
        tot, full, rej = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
        self._ck(self.lib.asb_panel_read_commit(self.h, ptr(words10), ctypes.byref(tot), ctypes.byref(full), ctypes.byref(rej)))
        return tot.value, full.value, bool(rej.value)

    def panel_set_coop(self, on):
        """Switches the co-resident panel kernel on / off; returns the previous setting."""
        return int(self.lib.asb_panel_set_coop(self.h, int(bool(on))))

    def panel_commit(self, k0, kept):
        self._ck(self.lib.asb_panel_commit(self.h, int(k0), int(kept)))

    def panel_refresh(self, k):
        e, g = ctypes.c_double(), ctypes.c_int64()
        self._ck(self.lib.asb_panel_refresh(self.h, int(k), ctypes.byref(e), ctypes.byref(g)))
        return e.value, g.value

    def deflate_stats(self):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.lib.asb_deflate_stats(self.h, ctypes.byref(a), ctypes.byref(b)))
        c, d = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.lib.asb_deflate_spec_stats(self.h, ctypes.byref(c), ctypes.byref(d)))
        e = ctypes.c_int64()
        self._ck(self.lib.asb_deflate_energy_passes(self.h, ctypes.byref(e)))
        f = ctypes.c_int64()
        self._ck(self.lib.asb_deflate_coop_fallbacks(self.h, ctypes.byref(f)))
        g = ctypes.c_int64()
        self._ck(self.lib.asb_deflate_guessed_panels(self.h, ctypes.byref(g)))
        h, i = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.lib.asb_deflate_sketch_stats(self.h, ctypes.byref(h), ctypes.byref(i)))
        j = ctypes.c_int64()
        self._ck(self.lib.asb_deflate_switch_stats(self.h, ctypes.byref(j)))
        return dict(panels=a.value, refreshes=b.value, unproven_tried=c.value, unproven_kept=d.value, energy_passes=e.value,
                    coop_fallbacks=f.value, guessed_panels=g.value, sketch_runs=h.value, sketch_reads=i.value,
                    residual_switch_at=j.value)

    def project_switch_residual(self, k):
        """The run leaves the projection mode at component k and continues in the residual loop (asb.h)."""
        self._ck(self.lib.asb_project_switch_residual(self.h, int(k)))
        self.mode = _lib.DEFLATE_RESIDUAL

    def download_residual(self):
        out = np.empty((self.F, self.n_loc, 3))
        self._ck(self.lib.asb_deflate_download_residual(self.h, ptr(out)))
        return out

    # ------------------------------------------------------------------ SPLOCS
    def splocs_begin(self):
        self._ck(self.lib.asb_splocs_begin(self.h))

    def splocs_gram(self, P_dev_ptr=None, M_dev_ptr=None, want_norm=False):
        nx = ctypes.c_double()
        self._ck(self.lib.asb_splocs_gram(self.h, ctypes.c_void_p(P_dev_ptr) if P_dev_ptr else None,
                                          ctypes.c_void_p(M_dev_ptr) if M_dev_ptr else None,
                                          ctypes.byref(nx) if want_norm else None))
        return nx.value if want_norm else None

    def splocs_weights(self, P_dev_ptr=None, M_dev_ptr=None):
        idx = np.empty(self.K, dtype=np.int64)
        val = np.empty(self.K)
        self._ck(self.lib.asb_splocs_weights(self.h, ctypes.c_void_p(P_dev_ptr) if P_dev_ptr else None,
                                             ctypes.c_void_p(M_dev_ptr) if M_dev_ptr else None, ptr(idx), ptr(val)))
        return idx, val

    def splocs_trace_begin(self, n_its):
        self._ck(self.lib.asb_splocs_trace_begin(self.h, int(n_its)))

    def splocs_objective_dev(self, it, P_dev_ptr=None, M_dev_ptr=None):
        self._ck(self.lib.asb_splocs_objective_dev(self.h, ctypes.c_void_p(P_dev_ptr) if P_dev_ptr else None,
                                                   ctypes.c_void_p(M_dev_ptr) if M_dev_ptr else None, int(it)))

    def splocs_trace(self, n_its):
        out = np.empty((int(n_its), 3))
        self._ck(self.lib.asb_splocs_trace(self.h, int(n_its), ptr(out)))
        return out

    def splocs_admm(self, Lambda_loc, rho, n_iter):
        Lambda_loc = np.ascontiguousarray(Lambda_loc, dtype=np.float64)
        assert Lambda_loc.shape == (self.K, self.n_loc)
        self._ck(self.lib.asb_splocs_admm(self.h, ptr(Lambda_loc), float(rho), int(n_iter)))

    def splocs_admm_fields(self, slots, lam, dmin, dmax, rho, n_iter):
        """ADMM step with Lambda built on the device from cached distance fields (geodesic_cache_add)."""
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        assert slots.shape == (self.K,)
        self._ck(self.lib.asb_splocs_admm_fields(self.h, ptr(slots), float(lam), float(dmin), float(dmax), float(rho), int(n_iter)))

    def splocs_objective(self, P_dev_ptr=None, M_dev_ptr=None):
        wp, gm, sp = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._ck(self.lib.asb_splocs_objective(self.h, ctypes.c_void_p(P_dev_ptr) if P_dev_ptr else None,
                                               ctypes.c_void_p(M_dev_ptr) if M_dev_ptr else None, ctypes.byref(wp),
                                               ctypes.byref(gm), ctypes.byref(sp)))
        return wp.value, gm.value, sp.value

    def splocs_results(self):
        C = np.empty((self.K, self.n_loc, 3))
        W = np.empty((self.F, self.K))
        self._ck(self.lib.asb_splocs_results(self.h, ptr(C), ptr(W)))
        return C, W

    # ------------------------------------------------------------------ post-processing
    # ------------------------------------------------------------------ device geodesics
    def geodesic_setup(self, heat, lap, grad, div, dense=False, coarse=None, slabs=None):
        """heat, lap, grad, div: scipy CSR matrices (float64).  dense: invert the two SPD systems explicitly on the
        device (asb_geodesic_dense_setup) instead of solving them by PCG per batch.  coarse = (agg, heat_c, lap_c): the
        aggregates and dense coarse operators of the PCG mode's two-level preconditioner."""
        keep = []

        def csr(m):
            m = m.tocsr()
            m.sum_duplicates()
            m.sort_indices()
            rp, ci, v = m.indptr.astype(np.int32), m.indices.astype(np.int32), np.ascontiguousarray(m.data, dtype=np.float64)
            keep.extend([rp, ci, v])
            return [rp.ctypes.data, ci.ctypes.data, v.ctypes.data]

        n, m3 = heat.shape[0], grad.shape[0]
        dh = np.ascontiguousarray(heat.diagonal(), dtype=np.float64)
        dl = np.ascontiguousarray(lap.diagonal(), dtype=np.float64)
        args = csr(heat) + csr(lap) + csr(grad) + csr(div) + [dh.ctypes.data, dl.ctypes.data]
        self._ck(self.lib.asb_geodesic_setup(self.h, int(n), int(m3), *args))
        self._geo_n = n
        self.geodesic_dense = bool(dense) or slabs is not None      # (the whole local step can stay on the device)
        if slabs is not None:
            # slab mode: order (permuted index -> vertex), ptr (slab boundaries); both matrices go over in the permuted numbering
            order, ptr_ = slabs
            order = np.ascontiguousarray(order, dtype=np.int64)
            inv = np.empty(n, dtype=np.int32)
            inv[order] = np.arange(n, dtype=np.int32)
            hp = heat.tocsr()[order][:, order]
            lp = lap.tocsr()[order][:, order]
            ptr32 = np.ascontiguousarray(ptr_, dtype=np.int32)
            self._ck(self.lib.asb_geodesic_bt_setup(self.h, int(ptr32.shape[0] - 1), ptr32.ctypes.data, inv.ctypes.data,
                                                    *(csr(hp) + csr(lp))))
        elif dense:
            self._ck(self.lib.asb_geodesic_dense_setup(self.h))
        elif coarse is not None:
            agg, heat_c, lap_c, omega = coarse
            agg = np.ascontiguousarray(agg, dtype=np.int32)
            nc = int(agg.max()) + 1
            order = np.argsort(agg, kind="stable").astype(np.int32)
            ptr_ = np.zeros(nc + 1, dtype=np.int32)
            np.cumsum(np.bincount(agg, minlength=nc), out=ptr_[1:])
            heat_c = np.ascontiguousarray(heat_c, dtype=np.float64)
            lap_c = np.ascontiguousarray(lap_c, dtype=np.float64)
            assert heat_c.shape == lap_c.shape == (nc, nc) and agg.shape == (n,)
            self._ck(self.lib.asb_geodesic_coarse_setup(self.h, nc, agg.ctypes.data, ptr_.ctypes.data, order.ctypes.data,
                                                        heat_c.ctypes.data, lap_c.ctypes.data, float(omega)))

    def apply_geodesic(self, k, dmin, dmax):
        self._ck(self.lib.asb_deflate_apply_geodesic(self.h, int(k), float(dmin), float(dmax)))

    GEODESIC_CACHE_SLOTS = 64 * 64

    def geodesic_cache_add(self, sources, tol=1e-13):
        """Solves the distance fields of ``sources`` and keeps them on the device; returns their cache slots."""
        src = np.ascontiguousarray(sources, dtype=np.int64)
        slots = []
        for b in range(0, src.shape[0], 64):
            part = src[b:b + 64]
            s0 = ctypes.c_int64()
            self._ck(self.lib.asb_geodesic_cache_add(self.h, ptr(part), int(part.shape[0]), float(tol), ctypes.addressof(s0)))
            slots.extend(range(s0.value, s0.value + part.shape[0]))
        return slots

    def geodesic_cache_clear(self):
        self._ck(self.lib.asb_geodesic_cache_clear(self.h))

    def geodesic_solve(self, sources, tol=1e-13):
        src = np.ascontiguousarray(sources, dtype=np.int64)
        out = np.empty((src.shape[0], self._geo_n))
        it = (ctypes.c_int * 2)()
        self._ck(self.lib.asb_geodesic_solve(self.h, ptr(src), int(src.shape[0]), float(tol), ptr(out), it))
        return out, (it[0], it[1])

    # ------------------------------------------------------------------ ingest
    def align_frames(self, frames, rigid=True):
        """Rigid Procrustes alignment of every frame to frame 0; returns (aligned (F,N,3) f64, T (F,4,4))."""
        fr = np.array(frames, dtype=np.float64, order="C", copy=True)
        T = np.empty((fr.shape[0], 4, 4))
        self._ck(self.lib.asb_align_frames(self.h, ptr(fr), fr.shape[0], fr.shape[1], int(bool(rigid)), ptr(T)))
        return fr, T

    # ------------------------------------------------------------------ POD / QR / DEIM (config 5)
    def pod_gram(self, G_dev_ptr=None, to_host=True):
        G = np.empty((self.F, self.F)) if to_host else None
        self._ck(self.lib.asb_pod_gram(self.h, ctypes.c_void_p(G_dev_ptr) if G_dev_ptr else None, ptr(G)))
        return G

    def pod_basis(self, V, sigma):
        V = np.ascontiguousarray(V, dtype=np.float64)
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        assert V.shape == (self.F, sigma.shape[0])
        self._ck(self.lib.asb_pod_basis(self.h, ptr(V), ptr(sigma), sigma.shape[0]))
        self.K = int(sigma.shape[0])

    def pod_project(self, B_dev_ptr=None, to_host=True):
        B = np.empty((self.K, self.F)) if to_host else None
        self._ck(self.lib.asb_pod_project(self.h, ctypes.c_void_p(B_dev_ptr) if B_dev_ptr else None, ptr(B)))
        return B

    def sym_tridiag(self, n, A_dev_ptr=None):
        """Householder tridiagonalisation of the n x n symmetric device matrix (default: the POD Gram matrix)."""
        d, e = np.empty(n), np.empty(max(n - 1, 0))
        self._ck(self.lib.asb_sym_tridiag(self.h, ctypes.c_void_p(A_dev_ptr) if A_dev_ptr else None, int(n), ptr(d),
                                          ptr(e) if n > 1 else ptr(np.empty(1))))
        return d, e

    def sym_backtransform(self, n, Z, A_dev_ptr=None):
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        assert Z.shape[0] == n
        V = np.empty_like(Z)
        self._ck(self.lib.asb_sym_backtransform(self.h, ctypes.c_void_p(A_dev_ptr) if A_dev_ptr else None, int(n), ptr(Z),
                                                int(Z.shape[1]), ptr(V)))
        return V

    def sym_eig_topk(self, n, k, A_dev_ptr=None, want_vectors=False):
        """All eigenvalues (descending) of the n x n symmetric device matrix (default: the POD Gram matrix) and its k
        leading eigenvectors, entirely on the device; the vectors stay there for ``pod_basis_dev``."""
        lam = np.empty(n)
        V = np.empty((n, k)) if want_vectors else None
        bad = ctypes.c_int64()
        self._ck(self.lib.asb_sym_eig_topk(self.h, ctypes.c_void_p(A_dev_ptr) if A_dev_ptr else None, int(n), int(k), ptr(lam),
                                           ptr(V), ctypes.byref(bad)))
        return (lam, V, bad.value) if want_vectors else (lam, bad.value)

    def pod_basis_dev(self, K):
        self._ck(self.lib.asb_pod_basis_dev(self.h, int(K)))
        self.K = int(K)

    def pod_slices(self, p, K):
        self._ck(self.lib.asb_pod_slices(self.h, int(p), int(K)))
        self.K = int(K)

    def pod_rotate(self, B_dev_ptr=None):
        S = np.empty(self.K)
        self._ck(self.lib.asb_pod_rotate(self.h, ctypes.c_void_p(B_dev_ptr) if B_dev_ptr else None, ptr(S)))
        return S

    def pod_deflate_begin(self, keep, B_dev_ptr=None):
        self._ck(self.lib.asb_pod_deflate_begin(self.h, ctypes.c_void_p(B_dev_ptr) if B_dev_ptr else None, int(keep)))

    def pod_deflate_end(self, last, K_total):
        self._ck(self.lib.asb_pod_deflate_end(self.h, int(last)))
        self.K = int(K_total)

    def pod_power(self, B_dev_ptr=None):
        self._ck(self.lib.asb_pod_power(self.h, ctypes.c_void_p(B_dev_ptr) if B_dev_ptr else None))

    def qr_apply_joint(self, G_dev_ptr=None):
        self._ck(self.lib.asb_qr_apply_joint(self.h, ctypes.c_void_p(G_dev_ptr) if G_dev_ptr else None))

    # host-array probes of the small dense device solvers (tests)
    def test_tridiag_eig(self, d, e, k):
        d = np.ascontiguousarray(d, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        n = d.shape[0]
        lam, Z, bad = np.empty(n), np.empty((n, max(k, 1))), ctypes.c_int64()
        self._ck(self.lib.asb_test_tridiag_eig(self.h, ptr(d), ptr(e) if n > 1 else ptr(np.zeros(1)), n, int(k), ptr(lam),
                                               ptr(Z), ctypes.byref(bad)))
        return lam, Z[:, :k], bad.value

    def test_jacobi_rows(self, A, want_u=True):
        A = np.ascontiguousarray(A, dtype=np.float64)
        nv, m = A.shape
        U = np.empty((nv, nv)) if want_u else None
        sig, sw = np.empty(nv), ctypes.c_int64()
        self._ck(self.lib.asb_test_jacobi_rows(self.h, ptr(A), nv, m, ptr(U), ptr(sig), ctypes.byref(sw)))
        return U, sig, sw.value

    def test_chol_tinv(self, G):
        G = np.ascontiguousarray(G, dtype=np.float64)
        Tt = np.empty_like(G)
        self._ck(self.lib.asb_test_chol_tinv(self.h, ptr(G), G.shape[0], ptr(Tt)))
        return Tt

    def snapshots_affine(self, inv_scale, add_mean, rowscale_loc=None):
        if rowscale_loc is not None:
            rowscale_loc = np.ascontiguousarray(rowscale_loc, dtype=np.float64)
        self._ck(self.lib.asb_snapshots_affine(self.h, float(inv_scale), int(bool(add_mean)), ptr(rowscale_loc)))

    def qr_apply(self, G_dev_ptr=None):
        self._ck(self.lib.asb_qr_apply(self.h, ctypes.c_void_p(G_dev_ptr) if G_dev_ptr else None))

    def deim_step(self, k, coef=None):
        if coef is not None:
            coef = np.ascontiguousarray(coef, dtype=np.float64)
            assert coef.shape == (3, k)
        i, v = ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.asb_deim_step(self.h, int(k), ptr(coef), ctypes.byref(i), ctypes.byref(v)))
        return i.value, v.value

    def deim_run(self):
        """The whole DEIM loop on the device (single rank): (Pt (K,), maxabs (K,), solve_failed)."""
        Pt, ma, bad = np.empty(self.K, dtype=np.int64), np.empty(self.K), ctypes.c_int()
        self._ck(self.lib.asb_deim_run(self.h, Pt.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ptr(ma), ctypes.byref(bad)))
        return Pt, ma, bad.value

    def deim_block_step(self, k, p, coef, group):
        """Residual of basis block k (p vectors) and its arg-max over rows (group = 1) or constraints (group = p):
        (index, energy, largest |r|)."""
        if coef is not None:
            coef = np.ascontiguousarray(coef, dtype=np.float64)
            assert coef.shape == (3, k * p, p)
        am, idx, val = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.asb_deim_block_residual(self.h, int(k), int(p), ptr(coef), ctypes.byref(am)))
        self._ck(self.lib.asb_energy_block_argmax(self.h, int(group), ctypes.byref(idx), ctypes.byref(val)))
        return idx.value, val.value, am.value

    def st_upload(self, St):
        """The sparse differential operator S^T (scipy sparse, |V| x e p) as CSR on the device."""
        St = St.tocsr()
        St.sort_indices()
        indptr = np.ascontiguousarray(St.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(St.indices, dtype=np.int64)
        data = np.ascontiguousarray(St.data, dtype=np.float64)
        self._ck(self.lib.asb_st_upload(self.h, int(St.shape[0]), int(St.shape[1]), int(data.shape[0]), ptr(indptr),
                                        ptr(indices), ptr(data)))

    def st_residual_argmax(self):
        v, val = ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.asb_st_residual_argmax(self.h, ctypes.byref(v), ctypes.byref(val)))
        return v.value, val.value

    def residual_norm2(self):
        out = ctypes.c_double()
        self._ck(self.lib.asb_deflate_residual_norm2(self.h, ctypes.byref(out)))
        return out.value

    def deim_block_step_st(self, k, p, coef):
        """Residual of basis block k mapped to position space by S^T: (vertex arg-max, its energy, largest |r|)."""
        if coef is not None:
            coef = np.ascontiguousarray(coef, dtype=np.float64)
            assert coef.shape == (3, k * p, p)
        am, v, val = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        self._ck(self.lib.asb_deim_block_residual_st(self.h, int(k), int(p), ptr(coef), ctypes.byref(am), ctypes.byref(v),
                                                     ctypes.byref(val)))
        return v.value, val.value, am.value

    def deim_row(self, gidx):
        row = np.empty((self.K, 3))
        rc = self.lib.asb_deim_row(self.h, int(gidx), ptr(row))
        if rc == 1:
            return None
        self._ck(rc)
        return row

    def components_stream(self, enable=True):
        """Overlapped download of the basis into pinned host memory (asb.h: asb_components_stream_into).  The buffer belongs
        to THIS object and to the ndarrays ``components_pinned`` returns, not to the context: switching the stream off,
        closing the engine or starting another run never frees or overwrites memory a live ndarray looks at."""
        self._streaming = bool(enable)
        if not enable:
            self._ck(self.lib.asb_components_stream_into(self.h, None, 0))
            self._pin = None

    def components_pinned(self):
        """(K, n_loc, 3) ndarray over the pinned buffer the run streamed its basis into.  Safe to keep: the memory lives as
        long as the array (or any view of it) does, and a later run on this engine takes a fresh buffer while it is alive."""
        pin = getattr(self, "_pin", None)
        if not getattr(self, "_streaming", False) or pin is None:
            raise RuntimeError("components_pinned: components_stream(True) was not on when the run began")
        p = ctypes.c_void_p()
        self._ck(self.lib.asb_components_pinned(self.h, ctypes.byref(p)))
        assert p.value == pin.ptr
        return pin.view((int(self.K), int(self.n_loc), 3))

    def deflate_reserve(self, K_new):
        """Residual mode: room for K_new components in all, keeping what the run has produced (asb.h: asb_deflate_reserve)."""
        self._ck(self.lib.asb_deflate_reserve(self.h, int(K_new)))
        self.K = max(int(self.K), int(K_new))

    def results_comps(self):
        out = np.empty((self.K, self.n_loc, 3))
        self._ck(self.lib.asb_components_download(self.h, ptr(out)))
        return out

    def components_expand(self, coef):
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        assert coef.ndim == 3 and coef.shape[0] == 3
        out = np.empty((coef.shape[2], self.n_loc, 3))
        self._ck(self.lib.asb_components_expand(self.h, ptr(coef), coef.shape[1], coef.shape[2], ptr(out)))
        return out

    def components_truncate(self, K):
        self._ck(self.lib.asb_components_truncate(self.h, int(K)))
        self.K = int(K)

    def components_upload(self, comps_loc):
        comps_loc = np.ascontiguousarray(comps_loc, dtype=np.float64)
        assert comps_loc.ndim == 3 and comps_loc.shape[1:] == (self.n_loc, 3)
        self._ck(self.lib.asb_components_upload(self.h, ptr(comps_loc), comps_loc.shape[0]))
        self.K = int(comps_loc.shape[0])

    def orth_gram(self, G_dev_ptr=None):
        self._ck(self.lib.asb_orth_gram(self.h, ctypes.c_void_p(G_dev_ptr) if G_dev_ptr else None))

    def orth_apply(self, G_dev_ptr=None):
        sing = np.empty((3, self.K))
        self._ck(self.lib.asb_orth_apply(self.h, ctypes.c_void_p(G_dev_ptr) if G_dev_ptr else None, ptr(sing)))
        return sing

    def orth_gram_get(self):
        G = np.empty((3, self.K, self.K))
        self._ck(self.lib.asb_orth_gram_get(self.h, ptr(G)))
        return G

    def components_transform(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        assert T.shape == (3, self.K, self.K)
        self._ck(self.lib.asb_components_transform(self.h, ptr(T)))

    def orth_refine(self, G_dev_ptr=None):
        self._ck(self.lib.asb_orth_refine(self.h, ctypes.c_void_p(G_dev_ptr) if G_dev_ptr else None))

    def components_post(self, unscale, pre_scale_factor, invMassL_loc=None, download=True):
        """``download=False``: the basis stays on the device (a later step changes or reads it there): no 8 K n_loc 3 bytes
        through a pageable buffer (config 5: 307 MB, 25 ms)."""
        out = np.empty((self.K, self.n_loc, 3)) if download else None
        if invMassL_loc is not None:
            invMassL_loc = np.ascontiguousarray(invMassL_loc, dtype=np.float64)
        self._ck(self.lib.asb_components_post(self.h, int(bool(unscale)), float(pre_scale_factor), ptr(invMassL_loc),
                                              ptr(out)))
        return out
