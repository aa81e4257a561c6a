"""cProfile of the config-5 DEIM stage (where does the host time go)."""
import cProfile, contextlib, io, pstats, sys, types
sys.path.insert(0, '.')
import numpy as np
from animsnapbases_amd import constraintsComponents, nonlinearSnapshots
ep, F, K = 50000, 4000, 256
rng = np.random.default_rng(5)
r = 40
frames = 0.1 + np.tensordot(rng.normal(size=(F, r)) * (0.85 ** np.arange(r))[None], rng.normal(size=(r, ep, 3)), (1, 0))
frames += 1e-5 * rng.normal(size=(F, ep, 3))
param = types.SimpleNamespace(constProj_rest_shape="first", constProj_numFrames=0, constProj_p_size=1,
                              constProj_massWeight=False, constProj_standarize=True, constProj_orthogonal=False,
                              constProj_basis_type="pod_vectorized", deim_desired_num_components=K,
                              constProj_store_sing_val=False, constProj_output_directory=".", name="c5", constProj_name="v")
with contextlib.redirect_stdout(io.StringIO()):
    ns = nonlinearSnapshots(param, frames=frames)
    ns.config(); ns.snapshots_prepare()
    cc = constraintsComponents(param, ns)
    cc.config(); cc.compute_components_store_singvalues(); cc.post_process_components()
    pr = cProfile.Profile()
    pr.enable()
    cc.deim()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
