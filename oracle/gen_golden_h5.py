"""TEST INFRASTRUCTURE ONLY.  ``.h5`` fixtures WRITTEN BY THE UNMODIFIED REFERENCE, for the three places the hot path
touches the reference's container format:

    /opt/conda/bin/python3.9 oracle/gen_golden_h5.py        # the image's interpreter that has h5py (3.3.0, NumPy 1.26)

* ``tests/golden/ref_sequence.h5``   -- ``utils/process.py:69-94 convert_sequence_to_hdf5`` run on the ``.off`` frames of
  ``tests/golden/ingest_small_off.npz`` (gzip'd float32 ``verts``, ``tris``, attrs ``mean`` / ``scale``);
* ``tests/golden/ref_components.h5`` -- ``snapbases/posComponents.py:330-341 store_animations`` on the golden basis of
  ``tests/golden/pca_global_small.npz`` (``default``, ``tris``, ``comp%03d``);
* ``tests/golden/ref_read.npz``      -- what ``snapbases/posSnapshots.py:108-121 read`` returns for ref_sequence.h5 as
  train AND test file (verts float64, tris, frs, nVerts).
The files are data the reference produced; its code never travels.
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.ref_import import import_reference      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    import h5py      # must be the real one: this script is pointless under the stub
    ref = import_reference()
    import utils.process as rp
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as work:
        os.chdir(work)
        try:
            off = np.load(os.path.join(OUT, "ingest_small_off.npz"))
            os.makedirs("off")
            for name in off.files:
                with open(os.path.join("off", name + ".off"), "w") as fh:
                    fh.write(str(off[name]))
            with contextlib.redirect_stdout(io.StringIO()):
                rp.convert_sequence_to_hdf5(os.path.join(work, "off", "frame_*.off"), lambda f: rp.load_off(f, no_colors=True),
                                            os.path.join(work, "seq.h5"), 100, 1)
            shutil.copy("seq.h5", os.path.join(OUT, "ref_sequence.h5"))

            g = np.load(os.path.join(OUT, "pca_global_small.npz"))
            comp = object.__new__(ref["posComponents"])
            comp.output_components_file = "components.h5"
            comp.comps = g["comps_post"]
            comp.pos_snapshots = types.SimpleNamespace(verts=g["verts"].astype(float), tris=g["tris"])
            comp.store_animations(work)
            shutil.copy("components.h5", os.path.join(OUT, "ref_components.h5"))

            snap = object.__new__(ref["posSnapshots"])
            snap.input_animation_file = snap.input_test_animation_file = os.path.join(work, "seq.h5")
            with contextlib.redirect_stdout(io.StringIO()):
                snap.read()
            np.savez_compressed(os.path.join(OUT, "ref_read.npz"), verts=snap.verts, tris=snap.tris, frs=np.array(snap.frs),
                                nVerts=np.array(snap.nVerts), test_verts=snap.test_verts, test_tris=snap.test_tris)
        finally:
            os.chdir(cwd)
    for n in ("ref_sequence.h5", "ref_components.h5", "ref_read.npz"):
        print("wrote", n, os.path.getsize(os.path.join(OUT, n)), "bytes; h5py", h5py.__version__)


if __name__ == "__main__":
    main()
