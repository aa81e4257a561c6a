"""CPU: the ``.h5`` paths of the drop-in (rows a1, a13, f-2 of SURVEY.md 8) against files written by the unmodified
reference.  h5py is not installed for the image's main interpreter; the checks (tests/h5_check.py) run in-process
where it is, else in the image's second interpreter that has it (/opt/conda/bin/python3.9, h5py 3.3.0), else skip."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ALT = os.environ.get("ASB_H5_PYTHON", "/opt/conda/bin/python3.9")


def _alt_has_h5py():
    if not os.path.exists(ALT):
        return False
    return subprocess.run([ALT, "-W", "ignore", "-c", "import h5py, numpy, scipy"], capture_output=True).returncode == 0


def test_h5_paths_match_reference_written_files():
    try:
        import h5py      # noqa: F401
    except ImportError:
        if not _alt_has_h5py():
            pytest.skip("no interpreter with h5py on this machine")
        r = subprocess.run([ALT, "-W", "ignore", os.path.join(HERE, "h5_check.py")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("ok ") == 3, r.stdout
        return
    import h5_check
    h5_check.main()
