"""GPU (-m gpu): the small dense solvers that replaced host LAPACK between the device steps (csrc/asb_smalldense.hip):
tridiagonal bisection + inverse iteration vs LAPACK's tridiagonal solvers, one-sided Jacobi vs numpy's SVD, blocked
Cholesky + triangular inverse vs scipy; then the paths built on them: POD with a CLUSTERED spectrum at F >= 1000 against
LAPACK's SVD of the snapshot matrix itself (the reference's call, constraintsComponents.py:307), `orth` / `qr` with
K = 200 and 512 (the reference's configurations use 200 ... 1000) entirely on the device."""
import types

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from animsnapbases_amd import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _gram_like_tridiag(n, rank, noise, seed):
    """Tridiagonal form of a Gram-like matrix: geometric decay of `rank` leading singular values over a noise floor."""
    from scipy.linalg import hessenberg
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    s = np.concatenate([0.8 ** np.arange(rank), noise * (1 + rng.random(n - rank))])[:n]
    G = (Q * s ** 2) @ Q.T
    T = hessenberg(0.5 * (G + G.T))
    return np.diag(T).copy(), np.diag(T, 1).copy(), s


@pytest.mark.parametrize("n,rank,noise,k", [(3, 3, 1e-3, 3), (64, 10, 1e-4, 20), (65, 30, 1e-5, 65), (500, 40, 1e-6, 72),
                                            (1300, 60, 1e-5, 160)])
def test_tridiagonal_bisection_and_inverse_iteration(eng, n, rank, noise, k):
    from scipy.linalg import eigh_tridiagonal
    d, e, s = _gram_like_tridiag(n, rank, noise, seed=n)
    lam, Z, bad = eng.test_tridiag_eig(d, e, k)
    ref = eigh_tridiagonal(d, e, eigvals_only=True)[::-1]
    assert bad == 0
    assert np.abs(lam - ref).max() < 8e-16 * np.abs(ref).max() * max(1.0, np.log2(n))       # absolute accuracy eps |T|
    assert np.all(np.diff(lam) <= 0)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.allclose(np.linalg.norm(Z, axis=0), 1.0, atol=1e-13)
    res = np.abs(T @ Z - Z * lam[None, :k]).max()
    assert res < 1e-13 * np.abs(ref).max()                                                  # every vector: tiny residual
    # the well separated ones also vector by vector, the rest as a span (what the POD's Rayleigh-Ritz step needs)
    _, Zr = eigh_tridiagonal(d, e)
    Zr = Zr[:, ::-1]
    sep = min(int(np.sum(0.8 ** np.arange(rank) > 30 * noise)), k - 1)      # leading values clear of the noise floor
    for j in range(sep):
        gap = min(lam[j - 1] - lam[j] if j else np.inf, lam[j] - lam[j + 1])
        err = min(np.linalg.norm(Z[:, j] - Zr[:, j]), np.linalg.norm(Z[:, j] + Zr[:, j]))
        assert err < 50 * 2.2e-16 * np.abs(ref).max() / gap + 1e-12, (j, err, gap)
    Qz, _ = np.linalg.qr(Z)
    lead = Zr[:, :sep]
    assert np.linalg.norm(lead - Qz @ (Qz.T @ lead), axis=0).max() < 1e-7
    assert np.linalg.cond(Z) < 1e3                                                          # independent vectors


def test_tridiagonal_with_a_split_and_exact_multiplicity(eng):
    """e[j] = 0 splits the matrix; a repeated eigenvalue across the two blocks."""
    d = np.array([2.0, 1.0, 3.0, 2.0, 1.0, 5.0])
    e = np.array([0.5, 0.5, 0.0, 0.5, 0.25])
    lam, Z, bad = eng.test_tridiag_eig(d, e, 6)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    ref = np.linalg.eigvalsh(T)[::-1]
    assert np.abs(lam - ref).max() < 1e-14
    assert np.abs(T @ Z - Z * lam[None]).max() < 1e-13


@pytest.mark.parametrize("nv,m", [(1, 5), (2, 2), (7, 3), (9, 40), (64, 64), (129, 700), (288, 4000)])
def test_one_sided_jacobi_vs_numpy_svd(eng, nv, m):
    rng = np.random.default_rng(nv * 1000 + m)
    r = min(nv, m)
    A = rng.normal(size=(nv, m)) * (0.9 ** np.arange(nv))[:, None]
    U, sig, sweeps = eng.test_jacobi_rows(A)
    Ur, Sr, _ = np.linalg.svd(A, full_matrices=False)
    assert sweeps <= 20
    assert np.abs(sig[:r] - Sr).max() < 1e-13 * Sr[0]
    assert np.allclose(U.T @ U, np.eye(nv), atol=1e-12)
    for j in range(r):
        gap = min(Sr[j - 1] - Sr[j] if j else np.inf, Sr[j] - (Sr[j + 1] if j + 1 < r else 0.0))
        if gap > 1e-6 * Sr[0]:
            err = min(np.linalg.norm(U[:, j] - Ur[:, j]), np.linalg.norm(U[:, j] + Ur[:, j]))
            assert err < 1e-9, (j, err)
    # U^T A has orthogonal rows of norm sig
    B = U.T @ A
    Gm = B @ B.T
    assert np.abs(Gm - np.diag(np.diag(Gm))).max() < 1e-12 * Sr[0] ** 2


@pytest.mark.parametrize("K", [1, 5, 32, 33, 100, 200, 512, 1000])
def test_blocked_cholesky_inverse(eng, K):
    import scipy.linalg as sla
    rng = np.random.default_rng(K)
    M = rng.normal(size=(K + 20, K))
    G = M.T @ M
    Tt = eng.test_chol_tinv(G)
    L = sla.cholesky(G, lower=True)
    ref = sla.solve_triangular(L, np.eye(K), lower=True).T       # Tt[c][i] = (L^-1)[i][c]
    assert relerr(Tt, ref) < 1e-10
    assert np.abs(np.triu(Tt.T, 1)).max() == 0.0                  # L^-1 is lower triangular
    assert np.allclose(Tt.T @ G @ Tt, np.eye(K), atol=1e-9)
    with pytest.raises(Exception):
        eng.test_chol_tinv(-G)


def _cparam(K, orth, tmp):
    return types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=1,
                                 constProj_massWeight=False, constProj_standarize=True, constProj_orthogonal=orth,
                                 constProj_basis_type="pod_vectorized", deim_desired_num_components=K,
                                 constProj_store_sing_val=False, constProj_output_directory=str(tmp), name="t", constProj_name="v")


def _pod(frames, K, tmp, orth=False):
    from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
    param = _cparam(K, orth, tmp)
    ns = nonlinearSnapshots(param, frames=frames)
    ns.config()
    ns.snapshots_prepare()
    cc = constraintsComponents(param, ns)
    cc.config()
    cc.compute_components_store_singvalues()
    return ns, cc


@pytest.mark.parametrize("F,ep,K", [(1000, 700, 48), (1200, 500, 60)])
def test_pod_clustered_spectrum_vs_lapack_on_A(F, ep, K, tmp_path):
    """Singular values in tight clusters (pairs 1e-9 apart, a triple, a plateau of eight equal ones) between well
    separated ones, over a noise floor: every requested vector that LAPACK's SVD of A determines is matched to 1e-5
    (BASELINE.json's bar), clusters as subspaces; singular values to 1e-10."""
    rng = np.random.default_rng(F)
    M = 3 * ep
    r = 64
    s = 0.85 ** np.arange(r)
    s[5] = s[4] * (1 - 1e-9)                     # near-double
    s[11] = s[10] * (1 - 1e-9)
    s[20:23] = s[20]                             # exact triple
    s[30:38] = s[30]                             # plateau of eight
    Uo, _ = np.linalg.qr(rng.normal(size=(M, r)))
    Vo, _ = np.linalg.qr(rng.normal(size=(F, r)))
    A = (Uo * s) @ Vo.T + 1e-9 * rng.normal(size=(M, F))
    frames = np.ascontiguousarray(A.T).reshape(F, ep, 3)
    frames = np.concatenate([np.zeros((1, ep, 3)), frames])       # rest frame 0 = 0: standardisation only scales
    ns, cc = _pod(frames, K, tmp_path)
    X = ((frames - frames[0:1]) * ns.pre_scale_factor).reshape(F + 1, -1).T      # (3 ep, F + 1): what the reference factors
    Ur, Sr, _ = np.linalg.svd(X, full_matrices=False)
    assert relerr(cc.singular_values[:K], Sr[:K]) < 1e-10
    got = cc.comps.reshape(K, -1).T                                # columns = left singular vectors
    assert np.allclose(got.T @ got, np.eye(K), atol=1e-10)
    # groups of (numerically) equal singular values are compared as subspaces, the rest vector by vector
    k = 0
    while k < K:
        j = k + 1
        while j < K and Sr[k] - Sr[j] < 1e-6 * Sr[k]:
            j += 1
        if j >= K and j < len(Sr) and Sr[j - 1] - Sr[j] < 1e-6 * Sr[j - 1]:
            break                                                  # a cluster cut by K: not determined by either solver
        P = Ur[:, k:j]
        assert np.linalg.norm(got[:, k:j] - P @ (P.T @ got[:, k:j])) < 1e-5 * np.sqrt(j - k), (k, j)
        k = j


@pytest.mark.parametrize("K", [200, 512])
def test_orth_and_qr_large_K_on_the_device(K, tmp_path, monkeypatch):
    """q_orthogonal (scipy.linalg.orth per dimension, posComponents.py:284-287) and constProj_orthogonal (economic QR,
    constraintsComponents.py:431-435) with K = 200 / 512: K x K eigen-problem by one-sided Jacobi, Cholesky blocked --
    no host LAPACK in between (numpy.linalg.eigh / scipy cholesky are made to fail)."""
    import scipy.linalg as sla
    from scipy.linalg import orth
    from animsnapbases_amd import posComponents, posSnapshots
    rng = np.random.default_rng(K)
    F, N = K + 40, 1100
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    param = types.SimpleNamespace(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=True,
                                  q_support="global", vertPos_numComponents=K, store_vertPos_PCA_sing_val=False,
                                  vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, vertPos_rest_shape="first",
                                  name="t", vertPos_output_directory=str(tmp_path))
    snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False)
    comp = posComponents(param, snaps)
    comp.compute_components_store_singvalues()
    pre_orth = comp.comps / snaps.pre_scale_factor + snaps.mean[None]
    frames = 0.1 + rng.normal(size=(F, 1300, 3))
    ns, cc = _pod(frames, K, tmp_path, orth=True)
    raw = cc.comps.copy() / ns.pre_scale_factor + ns.mean[None]

    def boom(*a, **k):
        raise AssertionError("host LAPACK called between device steps")
    with monkeypatch.context() as mp:          # what the host branches for K > 128 used to call
        mp.setattr(np.linalg, "eigh", boom)
        mp.setattr(sla, "cholesky", boom)
        mp.setattr(sla, "solve_triangular", boom)
        comp.post_process_components()
        cc.post_process_components()
    for l in range(3):
        ref = orth(pre_orth[:, :, l].T).T
        got = comp.comps[:, :, l]
        sg = np.sign(np.sum(got * ref, axis=1))
        assert relerr(got * sg[:, None], ref) < 1e-6
        assert np.allclose(got @ got.T, np.eye(K), atol=1e-10)
        refq = sla.qr(raw[:, :, l].T, mode="economic")[0].T
        gq = cc.comps[:, :, l]
        sg = np.sign(np.sum(gq * refq, axis=1))
        assert relerr(gq * sg[:, None], refq) < 1e-8
        assert np.allclose(gq @ gq.T, np.eye(K), atol=1e-11)
