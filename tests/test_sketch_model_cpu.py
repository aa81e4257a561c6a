"""CPU: the two models DESIGN.md 4 rests its candidate heuristics on, checked against the exact greedy loop (the NumPy oracle of
posComponents.py:67-122) on small noise-like data -- documentation that runs.

  * isotropic tail: a replay in the rank-one sketch "constant-in-time direction" with the rest of every slab taken as isotropic
    predicts the share g of that direction the first components leave behind, g_1 = (1/3) / (|a|^2 + 1/3) for the first winner;
  * cross terms: what the static scores EV + g M miss is zero-mean with variance 4 alpha^2 beta^2 M_v EV_v / (3 F) per step.
"""
import numpy as np

from oracle import asb_oracle as orc


def _data(N=4000, F=600, seed=3):
    rng = np.random.default_rng(seed)
    verts = rng.uniform(-1, 1, size=(F, N, 3))
    X = orc.prepare_snapshots(verts, "first", True)["snapTensor"]          # (F, N, 3), rest shape "first", one global scale
    return X


def test_isotropic_tail_predicts_the_share_left_of_the_constant_direction():
    X = _data()
    F, N, _ = X.shape
    d = orc.extract_k_components(X, 4)
    W = d["weigs"]                                                           # (F, K)
    ones = np.ones(F) / np.sqrt(F)
    g_exact = [1.0 - sum((ones @ (W[:, j] / np.linalg.norm(W[:, j]))) ** 2 for j in range(k + 1)) for k in range(4)]
    # the replay in the rank-one sketch: z_vd = x_vd . 1/sqrt(F), tail_v = E_v - |z_v|^2, Gram = z z^T + (tail / 3) I
    rows = X.transpose(1, 2, 0)                                              # (N, 3, F)
    z = rows @ ones                                                          # (N, 3)
    E = (rows * rows).sum((1, 2))
    tail = E - (z * z).sum(1)
    v = int(np.argmax(E))
    assert v == int(d["idx"][0])
    lam, U = np.linalg.eigh(np.outer(z[v], z[v]) + tail[v] / 3 * np.eye(3))
    u, lam = U[:, -1], lam[-1]
    q = z[v] @ u                                                             # the winner's coordinate on the constant direction
    g_model = 1.0 - q * q / lam                                              # what a deflation by w leaves of that direction
    a2 = (z[v] ** 2).sum() / F / (tail[v] / (3 * F) * 3)                     # |a|^2 with the noise variance per entry scaled to 1/3 (U[-1,1))
    assert abs(g_model - g_exact[0]) < 0.01, (g_model, g_exact)
    assert abs(g_model - (1 / 3) / (a2 + 1 / 3)) < 0.02                      # the closed form of DESIGN.md 4
    assert g_exact[0] < 0.2 and g_exact[1] < g_exact[0]                      # a falling share, about a tenth after one component


def test_cross_terms_have_the_variance_the_confidence_bound_uses():
    X = _data(N=6000, F=500, seed=5)
    F, N, _ = X.shape
    d = orc.extract_k_components(X, 1)
    w = d["weigs"][:, 0]
    w = w / np.linalg.norm(w)
    rows = X.transpose(1, 2, 0)
    ones = np.ones(F) / np.sqrt(F)
    alpha = ones @ w                                                         # w = alpha 1 + beta n_w
    beta2 = 1.0 - alpha * alpha
    z = rows @ ones                                                          # coordinates on the constant direction
    c = rows @ w                                                             # exact coefficients of the first component
    drop = (c * c).sum(1)                                                    # exact energy drops
    M = (z * z).sum(1)
    E = (rows * rows).sum((1, 2))
    EV = E - M
    # model: drop = alpha^2 M + (cross term) + beta^2 (n . n_w)^2, cross term ~ N(0, 4 alpha^2 beta^2 M EV / (3 F))
    resid = drop - alpha * alpha * M - beta2 * EV / F
    pred_var = 4 * alpha * alpha * beta2 * M * EV / (3 * F)
    keep = np.ones(N, bool)
    keep[int(d["idx"][0])] = False                                           # the winner itself is not "another vertex"
    ratio = (resid[keep] ** 2).mean() / pred_var[keep].mean()
    assert 0.7 < ratio < 1.4, ratio
    # and it is the size of the spread among the leading vertices: what no static score can order
    top = np.sort(EV)[::-1]
    assert np.sqrt(pred_var[keep].mean()) > 0.2 * (top[10] - top[300])
