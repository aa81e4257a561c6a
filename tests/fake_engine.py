"""TEST DOUBLE (CPU) for ``animsnapbases_amd.engine.HipEngine`` -- lives in tests/ only.

It lets the world-size-2 ``gloo`` tests exercise the product's MULTI-RANK HOST LOGIC
(vertex partition, scalar all-reduces of the standardisation, per-component record
all-gather + winner selection, norm reduction, basis gather) without a GPU.  The per-shard
arithmetic here is NumPy written to the C-ABI contract of include/asb.h; it is test
infrastructure, never a product fallback (the product raises without the HIP library).
"""
import ctypes

import numpy as np
import scipy.linalg as sla


def _view(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))


class FakeEngine(object):
    device_exchange = False          # exchange records are CPU tensors

    def __init__(self):
        self.F = self.n_loc = self.v0 = self.N_glob = self.K = 0
        self.mode = 0

    # ---- snapshots
    def upload(self, X, v0, n_loc, massL=None):
        X = np.asarray(X, dtype=np.float64)
        self.F, self.N_glob = X.shape[0], X.shape[1]
        self.v0, self.n_loc = int(v0), int(n_loc)
        self.X = X[:, v0:v0 + n_loc, :].copy()
        if massL is not None:
            self.X *= massL[v0:v0 + n_loc][None, :, None]

    def center(self, code, subtract):
        self.mean = self.X[0].copy() if code == 0 else self.X.mean(axis=0)
        if subtract:
            self.X -= self.mean[None]
        return float(self.X.sum())

    def sqdev(self, mu):
        return float(((self.X - mu) ** 2).sum())

    def scale(self, a):
        self.X *= a

    def get_mean(self):
        return self.mean.copy()

    def download_snapshots(self):
        return self.X.copy()

    # ---- deflation (residual mode contract)
    def deflate_begin(self, K, local_support, mode=0):
        self.mode = int(mode)
        self.K, self.local = int(K), bool(local_support)
        if self.mode == 1:
            return self._project_begin()
        self.R = self.X.copy()
        self.W = np.zeros((K, self.F))
        self.C = np.zeros((K, self.n_loc, 3))
        self.sigma = np.zeros(K)
        self.wn2 = np.zeros(K)
        self.idx = np.zeros(K, dtype=np.int64)
        self.nr2 = np.zeros(K)

    def xchg_len(self):
        return 2 + 3 * self.F

    def _energy(self):
        return (self.R ** 2).sum(axis=(0, 2))

    def local_best(self, k, rec_ptr):
        rec = _view(rec_ptr, self.xchg_len())
        e = self._energy()
        b = int(np.argmax(e))
        rec[0] = e[b]
        rec[1:2].view(np.int64)[0] = self.v0 + b
        rec[2:] = self.R[:, b, :].T.ravel()

    def pick(self, k, recs_ptr=None, n_rec=0):
        if recs_ptr:
            recs = _view(recs_ptr, n_rec * self.xchg_len()).reshape(n_rec, -1)
            e = recs[:, 0]
            ids = recs[:, 1].copy().view(np.int64)
            best = max(range(n_rec), key=lambda r: (e[r], -ids[r]))
            slab = recs[best, 2:].reshape(3, self.F)
            gidx = int(ids[best])
        else:
            en = self._energy()
            b = int(np.argmax(en))
            slab, gidx = self.R[:, b, :].T, self.v0 + b
        lam, U = np.linalg.eigh(slab @ slab.T)
        u = U[:, -1]
        if u[np.argmax(np.abs(u))] < 0:
            u = -u
        w = slab.T @ u
        if self.local:
            def proj(x):
                x = np.maximum(0.0, x)
                return x if x.max() == 0 else x / x.max()
            p, q = proj(w), proj(-w)
            w = p if sla.norm(p) > sla.norm(q) else q
        self.W[k], self.sigma[k], self.wn2[k], self.idx[k] = w, np.sqrt(max(lam[-1], 0.0)), w @ w, gidx

    def get_pick(self, k):
        return int(self.idx[k]), float(self.sigma[k])

    def apply(self, k, s_loc=None):
        w = self.W[k]
        c = np.tensordot(w, self.R, (0, 0))
        if s_loc is not None:
            c = c * s_loc[:, None]
        c = c / self.wn2[k]
        self.C[k] = c
        self.R -= np.outer(w, c).reshape(self.R.shape)
        self.nr2[k] = (self.R ** 2).sum()

    def run_global(self, k0, k1):
        assert self.mode == 0
        for k in range(k0, k1):
            self.pick(k)
            self.apply(k)

    def results(self, want_comps=True, want_weigs=True):
        if self.mode == 1:
            self.nr2 = self.normX2 - np.cumsum(self.colsum)
        return dict(comps=self.C.copy() if want_comps else None, weigs=self.W.T.copy() if want_weigs else None,
                    idx=self.idx.copy(), sigma=self.sigma.copy(), normR2_local=self.nr2.copy())

    def download_residual(self):
        return self.R.copy()

    def components_post(self, unscale, psf, invMassL_loc=None, download=True):
        c = self.C.copy()
        if unscale:
            c = c / psf + self.mean[None]
        if invMassL_loc is not None:
            c = c * invMassL_loc[None, :, None]
        self.C = c
        return c.copy() if download else None

    def deflate_stats(self):
        return dict(panels=0, refreshes=0)

    def sync(self):
        pass

    # ---- projection-mode (panel) contract of include/asb.h, NumPy emulation --------------------------------
    NBINS = 2048
    M_TARGET, M_CAP = 12, 24             # tiny on purpose: forces several panels and early panel ends

    def _project_begin(self):
        K = self.K
        self.W = np.zeros((K, self.F))
        self.C = np.zeros((K, self.n_loc, 3))
        self.sigma, self.wn2, self.colsum = np.zeros(K), np.zeros(K), np.zeros(K)
        self.idx = np.zeros(K, dtype=np.int64)
        self.E = (self.X ** 2).sum(axis=(0, 2))
        self.normX2, self.e0max = float(self.E.sum()), float(self.E.max())
        self.n_panels = self.n_refresh = 0

    def panel_scale(self, set_e0max=-1.0):
        out = (self.normX2, self.e0max)
        if set_e0max >= 0:
            self.e0max = float(set_e0max)
        return out

    def panel_capacity(self):
        return self.M_CAP

    def panel_row_len(self):
        return 3 * self.F

    def _hist_buf(self, hist_ptr):
        if not hist_ptr:                 # NULL: the context's own (local) histogram
            if not hasattr(self, "_own_hist"):
                self._own_hist = np.zeros(self.NBINS, dtype=np.int32)
            return self._own_hist
        return np.ctypeslib.as_array((ctypes.c_int32 * self.NBINS).from_address(hist_ptr))

    def panel_hist(self, level, hist_ptr=None):
        h = self._hist_buf(hist_ptr)
        if level == 1:                                        # biased-exponent histogram
            self.above = 0
            e = self.E[self.E >= 0]
            b = ((e.view(np.int64) >> 52) & 0x7FF).astype(np.int64)
        else:
            e = self.E[(self.E >= self.lo) & (self.E < self.hi)]
            scale = self.NBINS / (self.hi - self.lo) if self.hi > self.lo else 0.0
            b = np.minimum(((e - self.lo) * scale).astype(np.int64), self.NBINS - 1)
        h[:] = np.bincount(b, minlength=self.NBINS).astype(np.int32)[:self.NBINS]

    def panel_target(self):
        return self.M_TARGET

    def panel_top_energies(self, out_ptr, cap):
        out = _view(out_ptr, cap + 1)
        out[:] = -1.0
        e = self.E[self.E > self.tau][:cap]
        out[:e.size] = e[::-1]                      # any order: the device export is unordered too
        out[cap] = self.tau

    GLOBAL_TAU_ON_DEVICE = True          # False: behave like ASB_ERR_LIMIT (the driver then selects with torch)

    def panel_global_tau(self, tab_ptr, world, cap):
        if not self.GLOBAL_TAU_ON_DEVICE:
            return None
        tab = _view(tab_ptr, world * (cap + 1)).reshape(world, cap + 1)
        exported = np.sort(tab[:, :cap].reshape(-1))[::-1]
        kth = max(exported[self.M_TARGET], 0.0) if exported.size > self.M_TARGET else 0.0
        self.tau = float(max(kth, tab[:, cap].max()))
        return (tab[:, :cap] > self.tau).sum(axis=1).astype(np.int64)

    def panel_set_tau(self, tau_ptr):
        self.tau = float(_view(tau_ptr, 1)[0])

    def panel_tau(self, level, hist_ptr=None):
        h = self._hist_buf(hist_ptr)
        acc, b = self.above, self.NBINS - 1
        while b >= 0 and acc + h[b] < self.M_TARGET:
            acc += int(h[b])
            b -= 1
        if b < 0:
            self.tau = -1.0 if level == 1 else np.nextafter(self.lo, -np.inf)
            if level == 1:
                self.lo = self.hi = 0.0
                self.above = acc
            return
        if level == 1:
            self.lo = 0.0 if b == 0 else float(np.ldexp(1.0, b - 1023))
            self.hi = float(np.ldexp(1.0, b - 1022))
            self.above, self.tau = acc, self.lo
        else:
            width = (self.hi - self.lo) / self.NBINS
            e_lo = self.lo + b * width
            e_hi = self.hi if b == self.NBINS - 1 else self.lo + (b + 1) * width
            tau = e_hi if (acc + h[b] > self.M_CAP and acc > 0) else e_lo
            self.tau = np.nextafter(tau, -np.inf)

    def _residual_rows(self, vloc, k):
        R = self.X[:, vloc, :].copy()                       # (F, m, 3)
        for j in range(k):
            R -= self.W[j][:, None, None] * self.C[j][vloc][None]
        return R

    def panel_select(self, k, rows_ptr, idx_ptr, forced_gidx=-1, global_all=False, want_counts=True):
        if forced_gidx >= 0:
            vloc = np.array([forced_gidx - self.v0]) if self.v0 <= forced_gidx < self.v0 + self.n_loc else np.zeros(0, np.int64)
        elif global_all:
            vloc = np.arange(self.n_loc)
        else:
            vloc = np.nonzero(self.E > self.tau)[0]
        overflow = vloc.size > self.M_CAP
        vloc = vloc[:self.M_CAP]
        rows = _view(rows_ptr, self.M_CAP * 3 * self.F).reshape(self.M_CAP, 3, self.F)
        ids = np.ctypeslib.as_array((ctypes.c_int64 * self.M_CAP).from_address(idx_ptr))
        if vloc.size:
            rows[:vloc.size] = self._residual_rows(vloc, k).transpose(1, 2, 0)
            ids[:vloc.size] = self.v0 + vloc
        return int(vloc.size), bool(overflow)

    def panel_assemble(self, rows_g_ptr, idx_g_ptr, counts, maxcount):
        world = len(counts)
        rows = _view(rows_g_ptr, world * maxcount * 3 * self.F).reshape(world, maxcount, 3, self.F)
        ids = np.ctypeslib.as_array((ctypes.c_int64 * (world * maxcount)).from_address(idx_g_ptr)).reshape(world, maxcount)
        self.candR = np.concatenate([rows[r, :counts[r]] for r in range(world)], axis=0).copy()
        self.cand_idx = np.concatenate([ids[r, :counts[r]] for r in range(world)]).copy()

    def panel_assemble_packed(self, packed_g_ptr, counts, maxcount):
        world, rl = len(counts), 3 * self.F
        stride = maxcount * (rl + 1)
        rows, ids = [], []
        for r in range(world):
            base = packed_g_ptr + 8 * r * stride
            rows.append(_view(base, maxcount * rl).reshape(maxcount, 3, self.F)[:counts[r]])
            ids.append(np.ctypeslib.as_array((ctypes.c_int64 * maxcount).from_address(base + 8 * maxcount * rl))[:counts[r]])
        self.candR = np.concatenate(rows, axis=0).copy()
        self.cand_idx = np.concatenate(ids).copy()

    def panel_run_spec(self, k0, steps, global_all, spec_max):
        done = self.panel_run(k0, steps, global_all, spec_max)
        return done, self.proven

    def panel_run(self, k0, steps, global_all=False, spec_max=0):
        theta = -np.inf if global_all else self.tau
        margin = 1e-11 * self.e0max
        done = 0
        self.n_panels += 1
        self.proven, self.e_win = -1, []
        for t in range(steps):
            k = k0 + t
            e = (self.candR ** 2).sum(axis=(1, 2))
            b = int(np.argmax(e))
            if not e[b] > theta + margin:
                if self.proven < 0:
                    self.proven = t
                if not e[b] > margin or t - self.proven >= spec_max:
                    break
            self.e_win.append(e[b])
            slab = self.candR[b]
            lam, U = np.linalg.eigh(slab @ slab.T)
            u = U[:, -1]
            if u[np.argmax(np.abs(u))] < 0:
                u = -u
            w = slab.T @ u
            self.W[k], self.sigma[k], self.wn2[k], self.idx[k] = w, np.sqrt(max(lam[-1], 0.0)), w @ w, self.cand_idx[b]
            c = self.candR @ w / self.wn2[k]                 # (m, 3)
            self.candR -= c[:, :, None] * w[None, None, :]
            done += 1
        if self.proven < 0:
            self.proven = done
        return done

    def panel_project_spec(self, k0, ncols, proven):
        """Coefficients of all ncols columns; energies untouched.  First unproven step a vertex outside the candidate
        set (E <= tau) would have won or tied within the margin."""
        margin = 1e-11 * self.e0max
        outside = ~(self.E > self.tau)
        e = self.E.copy()
        self._spec_loss = []
        first = ncols
        self.n_spec_tried = getattr(self, "n_spec_tried", 0) + ncols - proven
        self._spec_proven = proven
        for t in range(ncols):
            k = k0 + t
            y = np.tensordot(self.W[k], self.X, (0, 0))
            for j in range(k):
                y -= self.C[j] * (self.W[j] @ self.W[k])
            self.C[k] = y / self.wn2[k]
            if t >= proven and first == ncols and outside.any() and not (self.e_win[t] > e[outside].max() + margin):
                first = t
            loss = (self.C[k] ** 2).sum(axis=1) * self.wn2[k]
            self._spec_loss.append(loss)
            e = e - loss
        return first

    def panel_commit(self, k0, kept):
        self.n_spec_kept = getattr(self, "n_spec_kept", 0) + max(0, kept - self._spec_proven)
        for t in range(kept):
            self.E = np.maximum(self.E - self._spec_loss[t], 0.0)
            self.colsum[k0 + t] = self._spec_loss[t].sum()

    def panel_project(self, k0, ncols):
        for t in range(ncols):
            k = k0 + t
            y = np.tensordot(self.W[k], self.X, (0, 0))     # (n, 3)
            for j in range(k0):
                y -= self.C[j] * (self.W[j] @ self.W[k])
            self.C[k] = y / self.wn2[k]
            loss = (self.C[k] ** 2).sum(axis=1) * self.wn2[k]
            self.E = np.maximum(self.E - loss, 0.0)
            self.colsum[k] = loss.sum()

    def panel_refresh(self, k):
        self.n_refresh += 1
        R = self._residual_rows(np.arange(self.n_loc), k)
        self.E = (R ** 2).sum(axis=(0, 2))
        b = int(np.argmax(self.E))
        return float(self.E[b]), int(self.v0 + b)

    def deflate_stats_project(self):
        return dict(panels=self.n_panels, refreshes=self.n_refresh, unproven_tried=getattr(self, "n_spec_tried", 0),
                    unproven_kept=getattr(self, "n_spec_kept", 0))
