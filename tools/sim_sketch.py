"""CPU replay of the panel algorithm's READS of X on structured data, to design the sketch predictor (DESIGN.md 4,
"predicted candidates"): how many components a read commits when its candidates are (A) the largest energies at the
start of the read (what rounds 1-2 did) or (B) the vertices a greedy run in SKETCH space comes close to selecting,
the sketch being the coefficient columns the previous read computed for its rejected steps.  NumPy only.

    python tools/sim_sketch.py lowrank|random|smooth [N] [F] [K]
"""
import sys
import time
import numpy as np

kind = sys.argv[1] if len(sys.argv) > 1 else "lowrank"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
F = int(sys.argv[3]) if len(sys.argv) > 3 else 256
K = int(sys.argv[4]) if len(sys.argv) > 4 else 128
M_CAND = int(sys.argv[5]) if len(sys.argv) > 5 else 768
STEPS = 64
import os
RMAX = int(os.environ.get("RMAX", "64"))
rng = np.random.default_rng(77)


def make():
    if kind == "random":
        X = rng.uniform(-1, 1, (3 * N, F))
    elif kind == "bumps":
        r = 50
        rest = rng.uniform(0, 1, (N, 3))
        modes = np.empty((r, 3 * N))
        for j in range(r):
            c = rest[rng.integers(N)]
            rad = 0.1 + 0.25 * rng.uniform()
            dvec = rng.standard_normal(3)
            modes[j] = (np.exp(-((rest - c) ** 2).sum(1) / rad ** 2)[:, None] * (dvec / np.linalg.norm(dvec))[None] * 0.1).ravel()
        coef = rng.standard_normal((F, r)) * 0.9 ** np.arange(r)[None]
        X = (coef @ modes).T + rng.standard_normal((3 * N, 1))
        X += 1e-4 * rng.standard_normal((3 * N, F))
    else:
        r = 50
        decay = 0.9 if kind == "lowrank" else 0.97
        if kind == "smooth":
            r = 200
        coef = rng.standard_normal((F, r)) * decay ** np.arange(r)[None]
        modes = 0.02 * rng.standard_normal((r, 3 * N))
        X = (coef @ modes).T + rng.standard_normal((3 * N, 1))
        X += 1e-4 * rng.standard_normal((3 * N, F))
    X -= X[:, :1].copy()
    X /= X.std()
    return np.ascontiguousarray(X)


X = make()
E0 = (X * X).sum(1).reshape(N, 3).sum(1)


def rows_of(idx):
    idx = np.asarray(idx)
    return (3 * idx[:, None] + np.arange(3)[None]).ravel()


def cand_greedy(Rc, steps):
    """greedy on the candidates' exact residual rows (M, 3, F); returns local winners, w (unit), winner energies"""
    Rc = Rc.copy()
    Ec = (Rc * Rc).sum((1, 2))
    win, Wl, et = [], [], []
    for _ in range(steps):
        m = int(np.argmax(Ec))
        w = np.linalg.svd(Rc[m], full_matrices=False)[2][0]
        c = Rc @ w                        # (M, 3)
        Rc -= c[:, :, None] * w[None, None, :]
        et.append(Ec[m])
        Ec = Ec - (c * c).sum(1)
        win.append(m)
        Wl.append(w)
    return win, np.array(Wl), np.array(et)


def sketch_greedy(Z, E, steps, iso=True):
    """Z (r, 3N) sketch of the residual, E (N) exact energies; returns closeness scores and predicted winners"""
    Z = Z.copy()
    E = E.copy()
    r = Z.shape[0]
    tail = np.maximum(E - (Z * Z).sum(0).reshape(N, 3).sum(1), 0.0)
    score = np.zeros(N)
    pred = []
    for _ in range(steps):
        v = int(np.argmax(E))
        score = np.maximum(score, E / E[v])
        pred.append(v)
        A = Z[:, 3 * v:3 * v + 3].copy()
        G = A.T @ A + (tail[v] / 3 if iso else 0.0) * np.eye(3)
        lam, U = np.linalg.eigh(G)
        u, lam = U[:, -1], lam[-1]
        if lam <= 0:
            break
        q = A @ u
        d = q @ Z                                # (3N)
        Ev = E[v]
        E = E - (d * d).reshape(N, 3).sum(1) / lam
        Z -= np.outer(q, d / lam)
        # the winner itself: R' = (I - u u^T) R on the 3-row side, sketch part and tail alike
        Z[:, 3 * v:3 * v + 3] = A - np.outer(q, u)
        E[v] = max(Ev - lam, 0.0)
        tail[v] *= 2.0 / 3.0
    return score, pred


def orth_rows(D):
    """orthonormal rows spanning the rows of D (in order: earlier rows keep their direction), numerically negligible ones dropped"""
    out = []
    for d in D:
        d = d.copy()
        for _ in range(2):
            for o in out:
                d -= (o @ d) * o
        nrm = np.linalg.norm(d)
        if nrm > 1e-6:
            out.append(d / nrm)
    return np.array(out) if out else np.zeros((0, D.shape[1]))


def run(method):
    k = 0
    W = np.zeros((0, F))
    C = np.zeros((0, 3 * N))
    E = E0.copy()
    reads = 0
    Zprev = None
    Dprev = None
    log = []
    while k < K:
        if method == "sketch2" and Dprev is not None and Dprev.shape[0] >= 4:
            # combined sketch: exact coordinates of the current residual on the kept-deflated old directions + new ones
            Zc = (X @ Dprev.T).T - (Dprev @ W.T) @ C if len(W) else (X @ Dprev.T).T
            score, pred = sketch_greedy(Zc, E, min(STEPS, K - k))
            cand = np.argpartition(-score, M_CAND)[:M_CAND]
        elif method == "energy" or Zprev is None or Zprev.shape[0] < 4:
            cand = np.argpartition(-E, M_CAND)[:M_CAND]
        else:
            score, pred = sketch_greedy(Zprev, E, min(STEPS, K - k))
            cand = np.argpartition(-score, M_CAND)[:M_CAND]
        rr = rows_of(cand)
        Rc = (X[rr] - C[:, rr].T @ W).reshape(len(cand), 3, F)
        steps = min(STEPS, K - k)
        win, Wl, et = cand_greedy(Rc, steps)
        Cl = (X @ Wl.T).T                         # the pass: (steps, 3N)
        reads += 1
        # check: first step at which a vertex outside the candidates has at least the winner's energy
        inside = np.zeros(N, bool)
        inside[cand] = True
        Et = E.copy()
        kept = steps
        for t in range(steps):
            if (Et[~inside] >= et[t]).any():
                kept = t
                break
            Et = Et - (Cl[t] ** 2).reshape(N, 3).sum(1)
        if kept == 0:
            # forced: exact first arg-max alone
            v = int(np.argmax(E))
            R = X[3 * v:3 * v + 3] - C[:, 3 * v:3 * v + 3].T @ W
            w = np.linalg.svd(R, full_matrices=False)[2][0]
            Wl, Cl, kept = w[None], (X @ w)[None], 1
            reads += 1
            Zprev = None
        else:
            Zprev = Cl[kept:kept + RMAX].copy() if kept < steps else None
        for t in range(kept):
            E = E - (Cl[t] ** 2).reshape(N, 3).sum(1)
        W = np.vstack([W, Wl[:kept]])
        C = np.vstack([C, Cl[:kept]])
        if method == "sketch2":
            # new leftover directions first, then the old ones with the kept directions projected out
            parts = [Wl[kept:]] if kept < len(Wl) else []
            if Dprev is not None:
                Wk = Wl[:kept]
                parts.append(Dprev - (Dprev @ Wk.T) @ Wk)
            Dprev = orth_rows(np.vstack(parts))[:RMAX] if parts else None
        k += kept
        log.append(kept)
    return reads, log


PASS_MS = {1: 0.87, 2: 1.0, 3: 1.25, 4: 1.41}


def cost_ms(sub, replay_steps=0):
    return 0.45 + 0.23 * sub + PASS_MS[sub] + 0.7 * replay_steps / 64.0


def run_short(steps_per_read):
    """persistent sketch (old directions deflated by the kept components, re-orthogonalised beside the new ones), reads of
    `steps_per_read` steps; candidates by the replay's scores.  Returns reads, kept per read and the modelled time."""
    k, reads, ms = 0, 0, 0.0
    W = np.zeros((0, F)); C = np.zeros((0, 3 * N)); E = E0.copy()
    D = None
    log = []
    while k < K:
        steps = min(steps_per_read, K - k)
        sub = (steps + 15) // 16
        if D is not None and D.shape[0] >= 4:
            Zc = (X @ D.T).T - (D @ W.T) @ C if len(W) else (X @ D.T).T
            score, _ = sketch_greedy(Zc, E, steps)
            cand = np.argpartition(-score, M_CAND)[:M_CAND]
            ms += cost_ms(sub, steps)
        else:
            cand = np.argpartition(-E, M_CAND)[:M_CAND]
            ms += cost_ms(sub)
        rr = rows_of(cand)
        Rc = (X[rr] - C[:, rr].T @ W).reshape(len(cand), 3, F)
        win, Wl, et = cand_greedy(Rc, steps)
        Cl = (X @ Wl.T).T
        reads += 1
        inside = np.zeros(N, bool); inside[cand] = True
        Et = E.copy(); kept = steps
        for t in range(steps):
            if (Et[~inside] >= et[t]).any():
                kept = t
                break
            Et = Et - (Cl[t] ** 2).reshape(N, 3).sum(1)
        if kept == 0:
            v = int(np.argmax(E))
            R = X[3 * v:3 * v + 3] - C[:, 3 * v:3 * v + 3].T @ W
            w = np.linalg.svd(R, full_matrices=False)[2][0]
            Wl, Cl, kept = w[None], (X @ w)[None], 1
            reads += 1
            ms += cost_ms(1)
        for t in range(kept):
            E = E - (Cl[t] ** 2).reshape(N, 3).sum(1)
        parts = [Wl[kept:]] if kept < len(Wl) else []
        if D is not None:
            Wk = Wl[:kept]
            parts.append(D - (D @ Wk.T) @ Wk)
        D = orth_rows(np.vstack(parts))[:RMAX] if parts else None
        W = np.vstack([W, Wl[:kept]]); C = np.vstack([C, Cl[:kept]])
        k += kept
        log.append(kept)
    return reads, log, ms


if os.environ.get("SHORT"):
    for spr in (16, 32, 64):
        t0 = time.time()
        reads, log, ms = run_short(spr)
        print("%s persistent sketch, %d-step reads: reads %d, modelled %.1f ms, kept %s (%.0f s)" % (kind, spr, reads, ms, log, time.time() - t0), flush=True)
    sys.exit(0)

for method in (os.environ.get("METHODS", "energy,sketch,sketch2").split(",")):
    t0 = time.time()
    reads, log = run(method)
    print("%s %s N=%d F=%d K=%d M=%d: reads %d  kept per read %s  (%.0f s)" % (kind, method, N, F, K, M_CAND, reads, log, time.time() - t0),
          flush=True)
