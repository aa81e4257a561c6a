"""Times the dense SPD inverse of the device geodesics (asb_dense_spd_inverse) at the size of bunny / armadillo, symmetric
form against the plain one (ASB_DENSE_SYM=0), and checks it against the identity."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from animsnapbases_amd import HipEngine
from animsnapbases_amd._lib import ptr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 14304
rng = np.random.default_rng(0)
B = rng.normal(size=(n, 64))
A = B @ B.T + np.diag(rng.uniform(1.0, 2.0, size=n) * 64)
e = HipEngine(0)
out = np.empty((n, n))
for rep in range(2):
    t0 = time.time()
    e._ck(e.lib.asb_test_spd_inverse(e.h, ptr(np.ascontiguousarray(A)), n, ptr(out)))
    print("n = %d: %.3f s including the host copies (%s)" % (n, time.time() - t0, os.environ.get("ASB_DENSE_SYM", "1")))
x = rng.normal(size=(n, 4))
print("residual |A (A^-1 x) - x| / |x| = %.2e, asymmetry %.2e" % (np.linalg.norm(A @ (out @ x) - x) / np.linalg.norm(x),
                                                                   np.abs(out - out.T).max()))
