import sys, types, contextlib, io
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from animsnapbases_amd import HipEngine, posComponents, posSnapshots
rng = np.random.default_rng(33)
F, N, K = 96, 4000, 20
verts = rng.uniform(-1, 1, size=(F, N, 3))
param = types.SimpleNamespace(vertPos_bases_type="PCA", q_standarize=True, q_massWeight=False, q_orthogonal=False,
                q_support="global", vertPos_numComponents=K, store_vertPos_PCA_sing_val=False,
                vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, vertPos_rest_shape="first", name="t", vertPos_output_directory=".")
which = sys.argv[1]
for stepwise in ((False, True) if which == "both" else (True,)):
    eng = HipEngine(0, stream=0) if stepwise else None
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_arrays(verts, None, "first", standarize=True, massWeight=False, engine=eng)
        comp = posComponents(param, snaps)
        comp.deflate_mode = "project"
        comp._stepwise_panels = stepwise
        comp.compute_components_store_singvalues()
    print("stepwise", stepwise, "ok", comp.selected_vertices[:5])
