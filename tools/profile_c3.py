import sys, time, types, contextlib, io, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from animsnapbases_amd import posComponents, posSnapshots
from oracle import asb_oracle as orc
rest, tris = orc.synth_mesh(87, 170, seed=1)
verts = orc.synth_snapshots(rest, 1000, rank=50, seed=1, kind="bumps")
param = types.SimpleNamespace(vertPos_bases_type="SPLOCS", vertPos_numComponents=64, q_support="local", store_vertPos_PCA_sing_val=False,
    vertPos_smooth_min_dist=0.1, vertPos_smooth_max_dist=0.25, q_standarize=True, q_massWeight=False, q_orthogonal=False,
    vertPos_output_directory=".", name="c3", splocs_max_itrs=20, splocs_admm_num_itrs=10, splocs_lambda=2.0, splocs_rho=10.0)
with contextlib.redirect_stdout(io.StringIO()):
    snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False)
    comp = posComponents(param, snaps)
    pr = cProfile.Profile(); pr.enable()
    comp.compute_components_store_singvalues()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
