"""CPU replay of the first READ of config 4 (64 greedy steps on U[-1,1) data, rest shape "first", standardised) for several
seeds: at which step does the first winner fall outside the guessed candidate union, for the round-2 grid of g values and for
finer ones?  (bench.py's end-to-end cycles use other seeds than the timed step: two of three have a rejection in read 1.)

    python tools/sim_first_read.py <seed> [<seed> ...]      # torch seeds as bench.py uses them need a GPU; these are NumPy seeds
"""
import sys
import numpy as np

N, F, K = 100000, 2000, 64


def replay(seed):
    import os
    cache = "/tmp/sim_first_read_%d.npz" % seed
    if os.path.exists(cache):
        d = np.load(cache)
        return d["E0"], d["EV"], d["win"].tolist(), d["g"].tolist()
    rng = np.random.default_rng(seed)
    X = np.empty((N * 3, F))
    for i in range(0, N * 3, 30000):
        X[i:i + 30000] = rng.uniform(-1, 1, (30000, F))
    X -= X[:, :1].copy()
    X /= X.std()
    E = (X * X).sum(1).reshape(N, 3).sum(1)
    E0 = E.copy()
    S = X.sum(1).reshape(N, 3)
    EV = E0 - (S * S).sum(1) / F
    W, C, win = [], [], []
    ones = np.ones(F) / np.sqrt(F)
    g = []
    for _ in range(K):
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
        g.append(1 - sum((ones @ ww) ** 2 for ww in W))
    np.savez(cache, E0=E0, EV=EV, win=np.array(win), g=np.array(g))
    return E0, EV, win, g


def top(score, m):
    return set(np.argpartition(-score, m)[:m].tolist())


BASE = ((0.0, 0, 400), (0.02, 0, 140), (0.05, 0, 140), (0.12, 0, 90), (0.3, 0, 60))
GRIDS = {
    "round 2 (5 x g, 400/140/140/90/60)": BASE,
    "+ h=1.5: 300": BASE + ((0.0, 1.5, 300),),
    "+ h=1.5: 500": BASE + ((0.0, 1.5, 500),),
    "+ h=1.0: 500": BASE + ((0.0, 1.0, 500),),
    "+ h=2.0: 500": BASE + ((0.0, 2.0, 500),),
    "+ h=1.5: 400, (g=.01,h=1.5): 200": BASE + ((0.0, 1.5, 400), (0.01, 1.5, 200)),
    "+ h=1: 300, h=2: 300": BASE + ((0.0, 1.0, 300), (0.0, 2.0, 300)),
}
for seed in [int(a) for a in sys.argv[1:]] or [0]:
    E0, EV, win, g = replay(seed)
    M = E0 - EV
    sig = 0.52 * np.sqrt(np.maximum(M, 0.0) * EV.mean() / F) * np.sqrt(3.0)      # std of the cross terms of the first components (model)
    print("seed %d: share of the constant direction left after steps 1,2,4,8,16,32,64: %s" % (seed, np.round([g[i] for i in (0, 1, 3, 7, 15, 31, 63)], 4)))
    for name, grid in GRIDS.items():
        union = top(E0, 64)
        for gq, hq, mq in grid:
            union |= top(EV + gq * M + hq * sig, mq)
        ok = [v in union for v in win]
        print("   %-36s union %4d  first winner outside: step %s  (winners outside in 64: %d)" % (name, len(union), ok.index(False) if False in ok else "none in 64", ok.count(False)))
