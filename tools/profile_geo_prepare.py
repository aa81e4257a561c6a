"""cProfile of the dense geodesic set-up on config 3's mesh (armadillo, 14 793 vertices): host assembly against device inverses."""
import os, sys, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from animsnapbases_amd import posSnapshots
from oracle import asb_oracle as orc
rest, tris, g = bench._fixture_mesh("c3_armadillo_splocs")
verts = orc.synth_snapshots(rest, 40, rank=5, noise=1e-4, seed=1, kind=str(g["kind"]))
for rep in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        snaps = posSnapshots.from_arrays(verts, tris, "first", standarize=True, massWeight=False)
        pr = cProfile.Profile(); pr.enable()
        snaps.compute_geodesic_distance.prepare()
        snaps._engine.sync()
        pr.disable()
    if rep == 1:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25)
        print("\n".join(l for l in s.getvalue().splitlines() if l.strip()))
    del snaps
