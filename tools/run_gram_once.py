"""One POD Gram (config-5 size) -- target for rocprofv3 counter passes on k_syrk_tn."""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from animsnapbases_amd import HipEngine
ep, F = 50000, 4000
rng = np.random.default_rng(5)
X = rng.normal(size=(F, ep, 3))
e = HipEngine(0)
e.upload(X, 0, ep)
for _ in range(3):
    e.pod_gram(to_host=False)
e.sync()
