"""CPU oracle for the animSnapBases snapshot-reduction hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``animsnapbases_amd`` never does (it fails loudly when its HIP library is missing).

This is a NumPy/SciPy float64 *restatement* of the reference algorithm (the reference is
pure Python on NumPy/SciPy, so NumPy is the faithful oracle language).  Every function
names the reference lines it follows (paths relative to the reference checkout).

Parity pin: the reference ships no tests / golden vectors.  The oracle
is pinned by running the *unmodified* reference in the build container
(``oracle/gen_golden.py`` through ``oracle/ref_import.py``) and committing its inputs and
outputs as ``tests/golden/*.npz``; ``tests/test_oracle_vs_golden.py`` checks this module
against them (index sequences equal, values <= 1e-12 relative).

Tensor convention (same as the reference): snapshots ``X`` are C-order ``(F, N, 3)``
float64; components ``C`` are ``(K, N, 3)``; weights ``W`` are ``(F, K)``.
"""
import struct

import numpy as np
import scipy.linalg as sla
from scipy import sparse
from scipy.sparse.linalg import splu


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _veclen(v):
    """utils/process.py:150-152."""
    return np.sqrt(np.sum(v ** 2, axis=-1))


def _normalized(v):
    """utils/process.py:154-156."""
    return v / _veclen(v)[..., None]


def project_weight(x):
    """snapbases/posComponents.py:52-58 -- clamp at 0, scale so that max == 1."""
    x = np.maximum(0.0, x)
    top = x.max()
    if top == 0:
        return x
    return x / top


def support_map(phi, min_dist, max_dist):
    """snapbases/posComponents.py:61-64 (the part after the geodesic solve)."""
    return (np.clip(phi, min_dist, max_dist) - min_dist) / (max_dist - min_dist)


def prox_l1l2(Lambda, x, beta):
    """snapbases/posComponents.py:252-256 -- group soft threshold over the xyz axis."""
    xlen = np.sqrt((x ** 2).sum(axis=-1))
    with np.errstate(divide="ignore"):
        shrink = np.maximum(0.0, 1 - beta * Lambda / xlen)
    return x * shrink[..., None]


# --------------------------------------------------------------------------------------
# snapshot preparation   (snapbases/posSnapshots.py)
# --------------------------------------------------------------------------------------
def factorize_masses(mass):
    """snapbases/posSnapshots.py:155-160.

    The reference takes the Cholesky factor and inverse of the dense N x N matrix
    ``diag(mass)``; for a diagonal matrix that is ``sqrt`` / reciprocal entry by entry.
    Returns (mass, massL, invMassL), each (N,).
    """
    mass = np.asarray(mass, dtype=np.float64)
    massL = np.sqrt(mass)
    return mass.copy(), massL, 1.0 / massL


def read_mass_bin(path, n_expected):
    """snapbases/posSnapshots.py:142-149 -- ``<i n><i m>`` then n little-endian doubles."""
    with open(path, "rb") as fh:
        n, _m = struct.unpack("<ii", fh.read(8))
        assert n == n_expected
        return np.frombuffer(fh.read(8 * n), dtype="<f8").astype(np.float64)


def prepare_snapshots(verts, rest_shape="first", standarize=True, massL=None):
    """snapbases/posSnapshots.py:64-105 and :163-172 (without file I/O and geodesics).

    verts (F,N,3) -> dict(snapTensor, mean, pre_scale_factor).
    """
    snap = np.array(verts, dtype=np.float64, copy=True)
    if massL is not None:
        assert snap.shape[1] == massL.shape[0]
        snap *= massL[:, None]
    if rest_shape == "first":
        mean = snap[0].copy()
    elif rest_shape == "average":
        mean = np.mean(snap, axis=0)
    else:
        raise SystemExit("Error! unknown rest shape: %s" % rest_shape)
    psf = 1
    if standarize:
        snap -= mean[None]
        psf = 1 / np.std(snap)
        snap *= psf
    return dict(snapTensor=snap, mean=mean, pre_scale_factor=psf)


# --------------------------------------------------------------------------------------
# heat-method geodesics   (utils/support.py:81-208)
# --------------------------------------------------------------------------------------
def mesh_laplacian(verts, tris):
    """utils/support.py:81-136 -- cotan Laplacian L (row sums 0) and lumped areas A."""
    n = len(verts)
    rows, cols, vals = [], [], []
    for a, b, c in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        va, vb, vc = tris[:, a], tris[:, b], tris[:, c]
        u = verts[vb] - verts[va]
        v = verts[vc] - verts[va]
        half_cot = 0.5 * (u * v).sum(axis=1) / _veclen(np.cross(u, v))
        rows += [vb, vc]
        cols += [vc, vb]
        vals += [half_cot, half_cot]
    L = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    L = (L - sparse.spdiags(L * np.ones(n), 0, n, n)).tocsr()
    e1 = verts[tris[:, 1]] - verts[tris[:, 0]]
    e2 = verts[tris[:, 2]] - verts[tris[:, 0]]
    third_area = 0.5 * _veclen(np.cross(e1, e2)) / 3
    area = np.zeros(n)
    for i in range(3):
        bc = np.bincount(tris[:, i].astype(int), third_area)
        area[:len(bc)] += bc
    return L, sparse.spdiags(area, 0, n, n)


class Geodesics(object):
    """utils/support.py:139-208 -- Crane et al. heat method, m = 10."""

    def __init__(self, verts, tris, m=10.0):
        verts = np.asarray(verts, dtype=np.float64)
        self.verts, self.tris = verts, tris
        e01 = verts[tris[:, 1]] - verts[tris[:, 0]]
        e12 = verts[tris[:, 2]] - verts[tris[:, 1]]
        e20 = verts[tris[:, 0]] - verts[tris[:, 2]]
        self.area = 0.5 * _veclen(np.cross(e01, e12))
        nrm = _normalized(np.cross(_normalized(e01), _normalized(e12)))
        self.n_x_e01 = np.cross(nrm, e01)
        self.n_x_e12 = np.cross(nrm, e12)
        self.n_x_e20 = np.cross(nrm, e20)
        h = np.mean([_veclen(e01), _veclen(e12), _veclen(e20)])
        t = m * h ** 2
        L, A = mesh_laplacian(verts, tris)
        self.solve_heat = splu((A - t * L).tocsc()).solve
        self.solve_poisson = splu(L.tocsc()).solve

    def __call__(self, idx):
        n = len(self.verts)
        tris, verts = self.tris, self.verts
        u0 = np.zeros(n)
        u0[idx] = 1.0
        u = self.solve_heat(u0).ravel()
        grad = (1 / (2 * self.area))[:, None] * (
            self.n_x_e01 * u[tris[:, 2]][:, None]
            + self.n_x_e12 * u[tris[:, 0]][:, None]
            + self.n_x_e20 * u[tris[:, 1]][:, None])
        Xf = -grad / _veclen(grad)[:, None]
        div = np.zeros(n)
        for a, b, c in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
            va, vb, vc = tris[:, a], tris[:, b], tris[:, c]
            e1 = verts[vb] - verts[va]
            e2 = verts[vc] - verts[va]
            eo = verts[vc] - verts[vb]
            cot1 = 1 / np.tan(np.arccos((_normalized(-e2) * _normalized(-eo)).sum(axis=1)))
            cot2 = 1 / np.tan(np.arccos((_normalized(-e1) * _normalized(eo)).sum(axis=1)))
            div += np.bincount(va.astype(int),
                               0.5 * (cot1 * (e1 * Xf).sum(axis=1) + cot2 * (e2 * Xf).sum(axis=1)),
                               minlength=n)
        phi = self.solve_poisson(div).ravel()
        phi -= phi.min()
        return phi


# --------------------------------------------------------------------------------------
# greedy deflation ("PCA")   (snapbases/posComponents.py:67-129)
# --------------------------------------------------------------------------------------
def extract_k_components(snapTensor, K, support="global", geodesics=None,
                         min_dist=None, max_dist=None):
    """snapbases/posComponents.py:69-122.

    Returns dict(comps (K,N,3), weigs (F,K), measures (K,3) = [k, sigma1, ||R||_F],
    idx (K,) int64 selected vertices, R final residual (F,N,3), smaps (K,N) or None).
    """
    R = snapTensor.copy()
    F = R.shape[0]
    C, W, meas, picked, smaps = [], [], [], [], []
    for k in range(K):
        energy = (R ** 2).sum(axis=2).sum(axis=0)              # :78-80
        idx = int(np.argmax(energy))
        _, sing, Vt = sla.svd(R[:, idx, :].reshape(F, -1).T, full_matrices=False)   # :83
        wk = sing[0] * Vt[0, :]                                 # :85
        if support == "local":                                  # :87-96
            pos = project_weight(wk)
            neg = project_weight(-wk)
            wk = pos if sla.norm(pos) > sla.norm(neg) else neg
            s = 1 - support_map(geodesics(idx), min_dist, max_dist)
            smaps.append(s)
            ck = (np.tensordot(wk, R, (0, 0)) * s[:, None]) / np.inner(wk, wk)      # :102-103
        else:
            ck = np.tensordot(wk, R, (0, 0)) / np.inner(wk, wk)                      # :105
        C.append(ck)
        W.append(wk)
        R -= np.outer(wk, ck).reshape(R.shape)                  # :111
        meas.append([k, sing[0], sla.norm(R)])                  # :113
        picked.append(idx)
    return dict(comps=np.array(C), weigs=np.array(W).T, measures=np.array(meas),
                idx=np.array(picked, dtype=np.int64), R=R,
                smaps=np.array(smaps) if smaps else None)


# --------------------------------------------------------------------------------------
# SPLOCS global optimisation   (snapbases/posComponents.py:132-189)
# --------------------------------------------------------------------------------------
def splocs_glob_optimization(snapTensor, comps, weigs, R, geodesics, min_dist, max_dist,
                             num_iters_max, num_admm_iterations, splocs_lambda, splocs_rho):
    """snapbases/posComponents.py:135-189.

    The reference discards C/W and only prints the trace; the oracle
    returns everything so that tests can compare.  Returns dict(C, W, Lambda, U,
    trace (its,2) = [energy, E_rms], idx (its,K) support-map centres).
    """
    K, N = comps.shape[0], comps.shape[1]
    F = snapTensor.shape[0]
    Lambda = np.empty((K, N))
    U = np.zeros((K, N, 3))
    C = comps.copy()
    W = weigs.copy()
    X = snapTensor.copy()
    R = R.copy()
    trace, centres = [], []
    for it in range(num_iters_max):
        Rflat = R.reshape(F, N * 3)                              # :143
        for k in range(K):                                       # :144-156
            Ck = C[k].ravel()
            nk = np.inner(Ck, Ck)
            if nk <= 1.e-8:
                W[:, k] = 0
                continue
            Rflat += np.outer(W[:, k], Ck)
            opt = np.dot(Rflat, Ck) / nk
            W[:, k] = project_weight(opt)
            Rflat -= np.outer(W[:, k], Ck)
        cen = []
        for k in range(K):                                       # :158-165
            idx = int((C[k] ** 2).sum(axis=1).argmax())
            cen.append(idx)
            Lambda[k] = splocs_lambda * support_map(geodesics(idx), min_dist, max_dist)
        centres.append(cen)
        Z = C.copy()                                             # :168
        G = np.dot(W.T, W)
        c = np.dot(W.T, X.reshape(F, -1))
        fac = sla.cho_factor(G + splocs_rho * np.eye(K))         # :172
        for _ in range(num_admm_iterations):                     # :175-178
            C = sla.cho_solve(fac, c + splocs_rho * (Z - U).reshape(c.shape)).reshape(C.shape)
            Z = prox_l1l2(Lambda, C + U, 1. / splocs_rho)
            U = U + C - Z
        C = Z                                                    # :181
        R = X - np.tensordot(W, C, (1, 0))                       # :183
        sparsity = np.sum(Lambda * np.sqrt((C ** 2).sum(axis=2)))
        e_rms = sla.norm(R) / np.sqrt(3 * N * F)
        energy = (R ** 2).sum() + sparsity
        trace.append([energy, e_rms])
    return dict(C=C, W=W, Lambda=Lambda, U=U, R=R, trace=np.array(trace),
                idx=np.array(centres, dtype=np.int64))


# --------------------------------------------------------------------------------------
# post-processing and diagnostics   (snapbases/posComponents.py:275-356)
# --------------------------------------------------------------------------------------
def post_process_components(comps, pre_scale_factor=None, mean=None, orthogonal=False, invMassL=None):
    """snapbases/posComponents.py:277-292.  ``pre_scale_factor``/``mean`` None <=> q_standarize False."""
    comps = comps.copy()
    if pre_scale_factor is not None:
        comps /= pre_scale_factor
        comps += mean[None]
    if orthogonal:
        for l in range(comps.shape[2]):
            comps[:, :, l] = sla.orth(comps[:, :, l].T).T
    if invMassL is not None:
        assert comps.shape[1] == invMassL.shape[0]
        comps *= invMassL[:, None]
    return comps


def bases_sing_vals(comps):
    """snapbases/posComponents.py:344-356 -- per-dimension normalised singular values (K,3)."""
    s = np.empty((comps.shape[0], 3))
    for i in range(3):
        sing = sla.svd(comps[:, :, i], full_matrices=False, compute_uv=False)
        s[:, i] = sing / sing.max()
    return s


def utmu(comps, mass):
    """snapbases/posComponents.py:305-313 -- U^T M U per dimension, (3,K,K)."""
    return np.array([np.dot(comps[:, :, l], comps[:, :, l].T * mass[:, None]) for l in range(3)])


def frobenius_error(f, g):
    """snapbases/posComponents.py:217-223."""
    return sla.norm(f - g)


def relative_error_per_component(f, g):
    """snapbases/posComponents.py:225-237."""
    return [sla.norm(f[:, :, i] - g[:, :, i]) / sla.norm(f[:, :, i]) for i in range(3)]


def max_pointwise_error(f, g):
    """snapbases/posComponents.py:239-249."""
    return np.max(np.abs(f - g)) / np.max(f)


def test_convergence(snapTensor, comps, weigs, start, end, step):
    """snapbases/posComponents.py:192-214 -- reconstruction errors for k = start..end."""
    fro, mx, rx, ry, rz = [], [], [], [], []
    for k in range(start, end + 1, step):
        rec = np.tensordot(weigs[:, :k], comps[:k], axes=([1], [0]))
        fro.append(frobenius_error(snapTensor, rec))
        rel = relative_error_per_component(snapTensor, rec)
        rx.append(rel[0]); ry.append(rel[1]); rz.append(rel[2])
        mx.append(max_pointwise_error(snapTensor, rec))
    return fro, mx, rx, ry, rz


# --------------------------------------------------------------------------------------
# storage   (utils/utils.py:14-38)
# --------------------------------------------------------------------------------------
def components_bin_bytes(comps):
    """utils/utils.py:26-35 -- the exact bytes of ``<name>F<F>K<K>.bin``.

    ``<i N><i 3K>`` then for d, for k, for i: ``<d comps[k,i,d]`` (column-major N x 3K,
    columns ordered x-block, y-block, z-block).
    """
    K, N, dim = comps.shape
    return struct.pack("<ii", N, dim * K) + np.ascontiguousarray(comps.transpose(2, 0, 1)).astype("<f8").tobytes()


def components_bin_name(prefix, F, K, col="K"):
    """utils/utils.py:27."""
    return prefix + "F" + str(F) + col + str(K) + ".bin"


def components_npy_name(prefix, F, K):
    """utils/utils.py:38 (np.save appends .npy)."""
    return prefix + str(F) + "K" + str(K) + ".npy"


# --------------------------------------------------------------------------------------
# constraint-projection snapshots: POD + DEIM (config #5)
# --------------------------------------------------------------------------------------
def prepare_nonlinear_snapshots(frames, rest_shape="first", standarize=True, massL=None):
    """snapbases/nonlinear_snapshots.py:74-96 and :268-288.  frames (F, ep, 3)."""
    snap = np.array(frames, dtype=np.float64, copy=True)
    if massL is not None:
        snap *= massL[:, None]
    mean, psf = None, 1
    if standarize:
        if rest_shape == "first":
            mean = snap[0].copy()
        elif rest_shape == "average":
            mean = np.mean(snap, axis=0)
        else:
            raise SystemExit("Error! unknown rest shape: %s" % rest_shape)
        snap -= mean[None]
        psf = 1 / np.std(snap)
        snap *= psf
    return dict(snapTensor=snap, mean=mean, pre_scale_factor=psf)


def pod_vectorized(snapTensor, desired_num_components):
    """snapbases/constraintsComponents.py:298-320.  Returns dict(comps (K,ep,3), S (F,))."""
    F = snapTensor.shape[0]
    A = snapTensor.reshape(F, -1).T
    U, S, _ = sla.svd(A, full_matrices=False)
    C = U.T.reshape((F, snapTensor.shape[1], -1))
    if desired_num_components < C.shape[0]:
        C = C[:desired_num_components]
    return dict(comps=np.array(C), S=S)


def pca_blocks(snapTensor, K, p):
    """snapbases/constraintsComponents.py:324-412 ('pca_blocks', global support): K times, the constraint whose p
    rows carry the most residual energy is chosen (:86-92) and its p rows are deflated one after the other, each by
    the rank-1 SVD of its 3 x F slab.  Returns dict(comps (K p, ep, 3), weigs (F, K p), measures (K, 3 + p),
    points (K,), blocks (K p,))."""
    R = snapTensor.copy()
    F, ep, _ = R.shape
    e = ep // p
    C, W, pts, blocks, meas = [], [], [], [], []
    for k in range(K):
        mag = (R ** 2).sum(axis=(0, 2)).reshape(e, p).sum(axis=1)
        idx = int(np.argmax(mag))
        pts.append(idx)
        sig = []
        for i in range(p):
            row = idx * p + i
            _, s, Vt = sla.svd(R[:, row, :].T, full_matrices=False)
            wk = s[0] * Vt[0]
            ck = np.tensordot(wk, R, (0, 0)) / np.inner(wk, wk)
            R -= np.outer(wk, ck).reshape(R.shape)
            blocks.append(row)
            C.append(ck)
            W.append(wk)
            sig.append(s[0])
        meas.append([k, idx, np.linalg.norm(R)] + sig)
    return dict(comps=np.array(C), weigs=np.array(W).T, measures=np.array(meas), points=np.array(pts, dtype=np.int64),
                blocks=np.array(blocks, dtype=np.int64), R=R)


def post_process_constraint_components(comps, snapTensor, pre_scale_factor=None, mean=None,
                                       orthogonal=False, invMassL=None):
    """snapbases/constraintsComponents.py:415-443.  Returns (comps, snapTensor) copies."""
    comps, snap = comps.copy(), snapTensor.copy()
    if pre_scale_factor is not None:
        comps /= pre_scale_factor
        comps += mean[None]
        snap /= pre_scale_factor
        snap += mean[None]
    if orthogonal:
        for l in range(comps.shape[2]):
            comps[:, :, l] = sla.qr(comps[:, :, l].T, mode="economic")[0].T
    if invMassL is not None:
        comps *= invMassL[:, None]
        snap *= invMassL[:, None]
    return comps, snap


def deim(comps, p=1):
    """snapbases/constraintsComponents.py:797-860.

    comps (K, ep, 3).  Returns dict(Pt (K,), alpha (K,), alpha_ranges (K,)).
    """
    K = comps.shape[0]
    bases = comps.swapaxes(0, 1)                  # (ep, K, 3)
    d = bases.shape[2]
    Pt, alphas, ranges = [], [], []
    V = None
    for k in range(K):
        vk = bases[:, k, :]
        if k == 0:
            r = vk
        else:
            c = np.empty(vk.shape)
            for i in range(d):
                c[:, i] = V[:, :, i] @ np.linalg.lstsq(V[Pt, :, i], vk[Pt, i], rcond=None)[0]
            r = c - vk
            if np.allclose(r, np.zeros(r.shape)):
                raise ArithmeticError("DEIM: zero residual at k=%d" % k)
        idx = int(np.argmax((r ** 2).sum(axis=1)))
        Pt.append(idx)
        alphas.append(idx // p)
        ranges.append(k + 1)
        V = vk[:, None, :] if k == 0 else np.concatenate((V, vk[:, None, :]), axis=1)
    return dict(Pt=np.array(Pt), alpha=np.array(alphas), alpha_ranges=np.array(ranges))


# --------------------------------------------------------------------------------------
# synthetic inputs shared by tests / bench / golden generation
# --------------------------------------------------------------------------------------
def synth_mesh(rings, segs, seed=0, bumps=0.25):
    """A closed genus-0 triangle mesh with ``rings*segs + 2`` vertices: a lat-long sphere
    deformed by a few smooth bumps so that edge lengths and areas are non-uniform.
    (rings=76, segs=188 gives 14 290 vertices / 28 576 faces -- the counts of data/bunny.obj.)
    """
    rng = np.random.default_rng(seed)
    th = np.pi * (np.arange(1, rings + 1) / (rings + 1))
    ph = 2 * np.pi * np.arange(segs) / segs
    T, P = np.meshgrid(th, ph, indexing="ij")
    pts = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], axis=-1).reshape(-1, 3)
    V = np.vstack([[0, 0, 1.0], pts, [0, 0, -1.0]])
    for _ in range(4):
        d = _normalized(rng.normal(size=3))
        V = V * (1 + bumps * np.exp(-3 * (1 - V @ d / _veclen(V)))[:, None] * rng.uniform(0.3, 1.0))
    V = V * np.array([0.45, 0.35, 0.4])
    tri = []
    top, bot = 0, rings * segs + 1

    def vid(r, s):
        return 1 + r * segs + (s % segs)

    for s in range(segs):
        tri.append((top, vid(0, s), vid(0, s + 1)))
        tri.append((bot, vid(rings - 1, s + 1), vid(rings - 1, s)))
    for r in range(rings - 1):
        for s in range(segs):
            a, b, c, d2 = vid(r, s), vid(r, s + 1), vid(r + 1, s), vid(r + 1, s + 1)
            tri.append((a, c, b))
            tri.append((b, c, d2))
    return V.astype(np.float64), np.array(tri, dtype=np.int64)


def synth_snapshots(rest, F, rank=10, noise=1e-4, seed=0, mode_scale=0.02, decay=0.7, kind="iid"):
    """``X = rest + coef(F x r) . modes(r x N x 3) + noise``; seeded, float64.

    kind="iid": modes ~ N(0, mode_scale^2) per entry.  kind="bumps": each mode is a smooth
    localised bump around a random vertex (what SPLOCS is designed to find).
    """
    rng = np.random.default_rng(seed)
    N = rest.shape[0]
    if kind == "iid":
        modes = rng.normal(scale=mode_scale, size=(rank, N, 3))
    else:
        extent = np.ptp(rest, axis=0).max()
        modes = np.empty((rank, N, 3))
        for j in range(rank):
            centre = rest[rng.integers(N)]
            radius = extent * rng.uniform(0.15, 0.4)
            bump = np.exp(-((rest - centre) ** 2).sum(axis=1) / radius ** 2)
            modes[j] = bump[:, None] * _normalized(rng.normal(size=3))[None] * (5 * mode_scale)
    coef = rng.normal(size=(F, rank)) * (decay ** np.arange(rank))[None, :]
    X = rest[None] + np.tensordot(coef, modes, (1, 0)) + noise * rng.normal(size=(F, N, 3))
    return X


def synth_uniform_snapshots(F, N, seed):
    """Config 4's input ("throughput on pure random"): U[-1,1) frames from ONE ``default_rng(seed).uniform``
    call, plus a triangle strip over the vertices -- global support never queries the geodesics, but the reference
    factorises the two heat-method systems eagerly (posSnapshots.py:96-99) and needs some triangle list to do so."""
    verts = np.random.default_rng(seed).uniform(-1.0, 1.0, size=(F, N, 3))
    i = np.arange(N - 2, dtype=np.int32)
    return verts, np.stack([i, i + 1, i + 2], axis=1)


def synth_constraint_frames(F, ep, rank, decay, noise, seed, chunk=250):
    """Config 5's input (low rank + noise): (F, ep, 3) = 0.1 + coef (F x r) . modes (r x 3 ep) + noise,
    coef[:, j] ~ N(0, decay^2j); the noise is drawn ``chunk`` frames at a time (part of the definition: it fixes the order
    in which the generator is consumed and bounds the peak memory to one copy of the tensor)."""
    rng = np.random.default_rng(seed)
    coef = rng.normal(size=(F, rank)) * (decay ** np.arange(rank))[None, :]
    modes = rng.normal(size=(rank, ep * 3))
    X = np.empty((F, ep * 3))
    for f0 in range(0, F, chunk):
        f1 = min(F, f0 + chunk)
        X[f0:f1] = 0.1 + coef[f0:f1] @ modes + noise * rng.normal(size=(f1 - f0, ep * 3))
    return X.reshape(F, ep, 3)


# --------------------------------------------------------------------------------------
# snapshot ingest   (utils/process.py)
# --------------------------------------------------------------------------------------
def find_rbm_procrustes(frompts, topts, rigid):
    """utils/process.py:210-234 -- rigid-body motion [R | t] (4x4) moving frompts onto topts."""
    t0, t1 = frompts.mean(0), topts.mean(0)
    M = np.dot((topts - t1).T, frompts - t0)
    U, _, Vt = np.linalg.svd(M)
    R = np.dot(U, Vt)
    if np.linalg.det(R) < 0:
        R *= -1
    T0 = np.eye(4)
    if rigid:
        T0[:3, :3] = R
    T0[:3, 3] = t1 - np.dot(R, t0)
    return T0


def rbm_transform(v, M):
    """utils/process.py:196-208 for a 4x4 M (homogenise, multiply, de-homogenise)."""
    v1 = np.concatenate([v, np.ones(v.shape[:-1] + (1,), dtype=v.dtype)], axis=-1).reshape(-1, 4)
    out = v1 @ M.T
    return (out[:, :3] / out[:, 3:4]).reshape(v.shape)


def align_frames(verts, rigid):
    """utils/process.py:241-246 -- every frame onto frame 0; float32 result like the reference."""
    v0 = verts[0]
    Ts = [find_rbm_procrustes(v, v0, rigid) for v in verts]
    return np.array([rbm_transform(v, M) for v, M in zip(verts, Ts)], np.float32), np.array(Ts)
