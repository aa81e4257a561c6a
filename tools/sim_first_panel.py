"""CPU replay of the first greedy steps of config 4 (100 000 x 2000, rest shape "first", standardised) to see which
candidate sets contain the first 16 winners -- what the guessed first selection (asb_project.hip:
panel_candidates) was designed on.  NumPy only; ~1.5 min and 10 GB per seed.

    python tools/sim_first_panel.py <seed> [<seed> ...]
"""
import sys
import numpy as np

N, F, K = 100000, 2000, 16


def replay(seed):
    rng = np.random.default_rng(seed)
    X = np.empty((N * 3, F))
    for i in range(0, N * 3, 30000):
        X[i:i + 30000] = rng.uniform(-1, 1, (30000, F))
    X -= X[:, :1].copy()           # rest shape "first"
    X /= X.std()                   # posSnapshots.standarize: one global scale
    E = (X * X).sum(1).reshape(N, 3).sum(1)
    E0 = E.copy()
    S = X.sum(1).reshape(N, 3)
    EV = E0 - (S * S).sum(1) / F   # energy without the constant-in-time direction
    W, C, win = [], [], []
    for _ in range(K):             # the reference loop in projection form (global support: the w_k are orthogonal)
        v = int(np.argmax(E))
        win.append(v)
        R = X[3 * v:3 * v + 3].copy()
        for w, c in zip(W, C):
            R -= np.outer(c[3 * v:3 * v + 3], w)
        w = np.linalg.svd(R, full_matrices=False)[2][0]
        c = X @ w
        W.append(w)
        C.append(c)
        E = E - (c * c).reshape(N, 3).sum(1)
    ones = np.ones(F) / np.sqrt(F)
    g = [1 - sum((ones @ w) ** 2 for w in W[:j]) for j in range(1, 6)]
    return E0, EV, win, g


def top(score, m):
    return set(np.argpartition(-score, m)[:m].tolist())


for seed in [int(a) for a in sys.argv[1:]] or [0]:
    E0, EV, win, g = replay(seed)
    M = E0 - EV
    union = top(E0, 64)
    for gq, mq in ((0.0, 400), (0.02, 140), (0.05, 140), (0.12, 90), (0.3, 60)):
        union |= top(EV + gq * M, mq)
    sets = {"largest 768 energies": top(E0, 768), "64 energies + 800 EV": top(E0, 64) | top(EV, 800),
            "the union used (%d)" % len(union): union}
    print("seed %d: share of the constant direction left after 1..5 components %s" % (seed, np.round(g, 4)))
    for name, cand in sets.items():
        ok = [v in cand for v in win]
        print("   %-26s first winner outside: step %s" % (name, ok.index(False) if False in ok else "none in 16"))
