"""CPU replay of the reads of X on structured data, round 4: which CANDIDATES and which SKETCH DIRECTIONS make a read commit most.
Variants on top of tools/sim_sketch.py's persistent-sketch loop (run_short):
  DIVERSE=f   a share f of the candidate slots goes to an energy-weighted random sample of ALL vertices (rows that span the
              dominant frame subspace of the whole residual, not only that of the few largest vertices)
  CONST=1     the persistent sketch always holds the constant-in-time direction (rest shape "first": every row carries its
              own offset; the noise phase of any data set starts like the random tensor of the headline)
  SPR=n       greedy steps per read
    python tools/sim_sketch2.py bumps|lowrank|smooth|random [N] [F] [K]
"""
import os, sys, time
import numpy as np

kind = sys.argv[1] if len(sys.argv) > 1 else "bumps"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
F = int(sys.argv[3]) if len(sys.argv) > 3 else 256
K = int(sys.argv[4]) if len(sys.argv) > 4 else 128
M_CAND = int(os.environ.get("M_CAND", "768"))
RMAX = int(os.environ.get("RMAX", "64"))
DIVERSE = float(os.environ.get("DIVERSE", "0"))
DIVERSE_PRED = float(os.environ.get("DIVERSE_PRED", str(DIVERSE)))      # the same for reads whose candidates the replay named
CONST = int(os.environ.get("CONST", "0"))
SPR = int(os.environ.get("SPR", "64"))
SEED = int(os.environ.get("SEED", "77"))
PERSIST = int(os.environ.get("PERSIST", "1"))      # 0: the sketch is the LAST read's rejected columns only (what round 3 ships)
F32 = int(os.environ.get("F32", "0"))              # 1: the replay in float32 (what the device kernel holds its sketch in)
E_SHARE = float(os.environ.get("E_SHARE", "0"))    # share of a predicted read's slots given to the largest energies (device: 1/3)
POW = float(os.environ.get("POW", "1"))            # the weighted sample draws vertex v with probability ~ E_v^POW
FIRST_DIV = int(os.environ.get("FIRST_DIV", "-1"))   # >= 0: the first read takes M_CAND by energy + FIRST_DIV weighted-random vertices (the device's guessed read)
ADAPT = int(os.environ.get("ADAPT", "0"))          # 1: SPR doubles (up to 64) after a read that kept everything, halves (down to 16) below a third
rng = np.random.default_rng(SEED)


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
    Rc = Rc.copy()
    Ec = (Rc * Rc).sum((1, 2))
    win, Wl, et = [], [], []
    for _ in range(steps):
        m = int(np.argmax(Ec))
        w = np.linalg.svd(Rc[m], full_matrices=False)[2][0]
        c = Rc @ w
        Rc -= c[:, :, None] * w[None, None, :]
        et.append(Ec[m])
        Ec = Ec - (c * c).sum(1)
        win.append(m)
        Wl.append(w)
    return win, np.array(Wl), np.array(et)


def sketch_greedy(Z, E, steps):
    Z = Z.astype(np.float32) if F32 else Z.copy()
    E = E.astype(np.float32) if F32 else E.copy()
    tail = np.maximum(E - (Z * Z).sum(0).reshape(N, 3).sum(1), 0.0)
    score = np.zeros(N)
    pred = []
    for _ in range(steps):
        v = int(np.argmax(E))
        score = np.maximum(score, E / E[v])
        pred.append(v)
        A = Z[:, 3 * v:3 * v + 3].copy()
        G = A.T @ A + (tail[v] / 3) * np.eye(3)
        lam, U = np.linalg.eigh(G)
        u, lam = U[:, -1], lam[-1]
        if lam <= 0:
            break
        q = A @ u
        d = q @ Z
        Ev = E[v]
        E = E - (d * d).reshape(N, 3).sum(1) / lam
        Z -= np.outer(q, d / lam)
        Z[:, 3 * v:3 * v + 3] = A - np.outer(q, u)
        E[v] = max(Ev - lam, 0.0)
        tail[v] *= 2.0 / 3.0
    return score, pred


def orth_rows(D):
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


PASS_MS = {1: 0.87, 2: 1.0, 3: 1.25, 4: 1.41}


def cost_ms(sub, replay_steps=0):
    return 0.45 + 0.23 * sub + PASS_MS[sub] + 0.7 * replay_steps / 64.0


def pick_candidates(score, E, m, share=None):
    """m slots: (1 - share) by the score (or the energies), share by an energy-weighted sample of everyone else"""
    nd = int(round((DIVERSE if share is None else share) * m))
    ne = int(round(E_SHARE * m)) if score is not E else 0
    top = np.argpartition(-score, m - nd - ne)[:m - nd - ne]
    if ne:
        Em = E.copy(); Em[top] = -1.0
        top = np.concatenate([top, np.argpartition(-Em, ne)[:ne]])
    if nd == 0:
        return top
    p = np.maximum(E, 0.0) ** POW
    p[top] = 0.0
    p /= p.sum()
    extra = rng.choice(N, size=nd, replace=False, p=p)
    return np.concatenate([top, extra])


def run(steps_per_read):
    k, reads, ms = 0, 0, 0.0
    W = np.zeros((0, F)); C = np.zeros((0, 3 * N)); E = E0.copy()
    D = None
    if CONST:
        D = np.full((1, F), 1.0 / np.sqrt(F))
    log, how = [], []
    while k < K:
        steps = min(steps_per_read, K - k)
        sub = (steps + 15) // 16
        if D is not None and D.shape[0] >= (1 if CONST else 4):
            Zc = (X @ D.T).T - (D @ W.T) @ C if len(W) else (X @ D.T).T
            score, _ = sketch_greedy(Zc, E, steps)
            cand = pick_candidates(score, E, M_CAND, DIVERSE_PRED)
            ms += cost_ms(sub, steps)
            how.append("s%d" % D.shape[0])
        elif k == 0 and FIRST_DIV >= 0:
            cand = pick_candidates(E, E, M_CAND + FIRST_DIV, FIRST_DIV / float(M_CAND + FIRST_DIV))
            ms += cost_ms(sub)
            how.append("g")
        else:
            cand = pick_candidates(E, E, M_CAND)
            ms += cost_ms(sub)
            how.append("e")
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
            Wl2, Cl2 = w[None], (X @ w)[None]
            reads += 1
            ms += cost_ms(1)
            # the rejected read's columns still are sketch directions (deflated by the forced component below)
            rej = Wl
            Wl, Cl, kept = Wl2, Cl2, 1
            parts = [rej - (rej @ Wl.T) @ Wl]
        else:
            parts = [Wl[kept:]] if kept < len(Wl) else []
        for t in range(kept):
            E = E - (Cl[t] ** 2).reshape(N, 3).sum(1)
        if D is not None and (PERSIST or (CONST and not parts)):
            Wk = Wl[:kept]
            parts.append(D - (D @ Wk.T) @ Wk)
        elif D is not None and CONST:
            one = np.full((1, F), 1.0 / np.sqrt(F))
            parts.append(one - (one @ W.T) @ W - (one @ Wl[:kept].T) @ Wl[:kept] if len(W) else one - (one @ Wl[:kept].T) @ Wl[:kept])
        D = orth_rows(np.vstack(parts))[:RMAX] if parts else None
        if ADAPT:
            if kept >= steps:
                steps_per_read = min(64, 2 * steps_per_read)
            elif kept * 3 < steps:
                steps_per_read = max(16, steps_per_read // 2)
        W = np.vstack([W, Wl[:kept]]); C = np.vstack([C, Cl[:kept]])
        k += kept
        log.append(kept)
    return reads, log, ms, how


t0 = time.time()
reads, log, ms, how = run(SPR)
print("%s POW=%.2f FIRST_DIV=%d E_SHARE=%.2f seed %d PERSIST=%d ADAPT=%d DIVERSE=%.2f/%.2f CONST=%d SPR=%d M=%d: reads %d, modelled %.1f ms, kept %s how %s (%.0f s)" %
      (kind, POW, FIRST_DIV, E_SHARE, SEED, PERSIST, ADAPT, DIVERSE, DIVERSE_PRED, CONST, SPR, M_CAND, reads, ms, log, how, time.time() - t0), flush=True)
