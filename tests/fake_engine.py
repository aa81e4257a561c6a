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
        assert mode == 0, "the test double only implements the residual-mode contract"
        self.K, self.local = int(K), bool(local_support)
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
        for k in range(k0, k1):
            self.pick(k)
            self.apply(k)

    def results(self, want_comps=True, want_weigs=True):
        return dict(comps=self.C.copy() if want_comps else None, weigs=self.W.T.copy() if want_weigs else None,
                    idx=self.idx.copy(), sigma=self.sigma.copy(), normR2_local=self.nr2.copy())

    def download_residual(self):
        return self.R.copy()

    def components_post(self, unscale, psf, invMassL_loc=None):
        c = self.C.copy()
        if unscale:
            c = c / psf + self.mean[None]
        if invMassL_loc is not None:
            c = c * invMassL_loc[None, :, None]
        self.C = c
        return c.copy()

    def deflate_stats(self):
        return dict(panels=0, refreshes=0)

    def sync(self):
        pass
