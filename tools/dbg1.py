import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from animsnapbases_amd import HipEngine
from oracle import asb_oracle as orc
from conftest import align_signs, relerr
rng = np.random.default_rng(21)
N,F,K=6000,120,24
X = np.tensordot(rng.normal(size=(F, 8)) * (0.7 ** np.arange(8)), rng.normal(size=(8, N, 3)), (1, 0)) + 1e-4 * rng.normal(size=(F, N, 3))
ref = orc.extract_k_components(X, K)
for mode in (0,1):
    e = HipEngine(0); e.upload(X,0,N); e.deflate_begin(K, False, mode); e.run_global(0,K); r=e.results(); print(mode, e.deflate_stats()); e.close()
    comps, weigs = align_signs(r["comps"], r["weigs"], ref["comps"])
    print(mode, ["%.1e"%relerr(comps[k], ref["comps"][k]) for k in range(K)])
