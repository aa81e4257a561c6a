"""Snapshot ingest against vectors produced by the unmodified reference
(utils/process.py, generated with the image's NumPy-1.26 interpreter: the reference's ingest code uses
ndarray.ptp / np.asfarray, which NumPy 2 removed)."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import asb_oracle as orc


def _write_offs(tmp_path):
    g = load_golden("ingest_small_off")
    for k, v in g.items():
        (tmp_path / (k + ".off")).write_text(str(v))
    return str(tmp_path / "frame_*.off")


def test_load_off_and_preprocess_match_reference(tmp_path, capsys):
    from animsnapbases_amd import process as P
    g = load_golden("ingest_small")
    pattern = _write_offs(tmp_path)
    files = sorted(__import__("glob").glob(pattern))
    P.sort_nicely(files)
    loaded = [P.load_off(f, no_colors=True) for f in files]
    assert np.array_equal(np.array([l[0] for l in loaded]), g["off_verts"])
    assert np.array_equal(loaded[0][1], g["off_tris"])
    verts = np.array([l[0] for l in loaded], np.float32)
    pv, pt, removed, pmean, pscale = P.preprocess_mesh_animation(verts, loaded[0][1])
    assert np.array_equal(pt, g["pre_tris"]) and np.array_equal(removed, g["pre_removed"])
    assert relerr(pv, g["pre_verts"]) < 1e-6 and relerr(pmean, g["pre_mean"]) < 1e-6
    assert abs(pscale - float(g["pre_scale"])) < 1e-6 * float(g["pre_scale"])
    assert removed.sum() == 4                       # the small disconnected component went away
    assert P.alphanum_key("z23a") == ["z", 23, "a"]


def test_convert_sequence_npz_roundtrip(tmp_path, capsys):
    from functools import partial
    from animsnapbases_amd import process as P
    from animsnapbases_amd.utils import read_animation
    g = load_golden("ingest_small")
    pattern = _write_offs(tmp_path)
    out = str(tmp_path / "anim.npz")
    P.convert_sequence_to_hdf5(pattern, partial(P.load_off, no_colors=True), out, 100, 1)
    verts, tris = read_animation(out)
    assert relerr(verts, g["pre_verts"]) < 1e-6 and np.array_equal(tris, g["pre_tris"])
    P.convert_sequence_to_hdf5(pattern, partial(P.load_off, no_colors=True), out, 3, 2)     # every 2nd file, 3 frames
    verts, _ = read_animation(out)
    assert verts.shape[0] == 3


@pytest.mark.parametrize("rigid", [True, False])
def test_oracle_alignment_matches_reference(rigid):
    g = load_golden("ingest_small")
    al, T = orc.align_frames(g["pre_verts"], rigid)
    assert relerr(T, g["T_rigid%d" % int(rigid)]) < 1e-5          # the reference works in float32
    assert relerr(al, g["aligned_rigid%d" % int(rigid)]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("rigid", [True, False])
def test_device_alignment_matches_reference(rigid, tmp_path, capsys):
    from animsnapbases_amd import process as P
    from animsnapbases_amd.utils import read_animation
    g = load_golden("ingest_small")
    al, T = P.align_frames(g["pre_verts"], rigid)
    assert al.dtype == np.float32
    assert relerr(T, g["T_rigid%d" % int(rigid)]) < 1e-5           # north-star float tolerance; the reference is f32
    assert relerr(al, g["aligned_rigid%d" % int(rigid)]) < 1e-5
    al64, T64 = orc.align_frames(g["pre_verts"].astype(np.float64), rigid)
    assert relerr(T, T64) < 1e-7                                   # float32 input data; f64 arithmetic on both sides
    M = P.find_rbm_procrustes(g["pre_verts"][3], g["pre_verts"][0], rigid)
    assert relerr(M, T[3]) < 1e-12
    if rigid:       # rotations are proper, frame 0 maps to itself
        for t in T:
            assert abs(np.linalg.det(t[:3, :3]) - 1) < 1e-9
        assert np.allclose(T[0], np.eye(4), atol=1e-9)
    # file-level align(): npz in, npz out
    src, dst = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(src, verts=g["pre_verts"], tris=g["pre_tris"])
    P.align(src, dst, rigid)
    v2, t2 = read_animation(dst)
    assert relerr(v2, g["aligned_rigid%d" % int(rigid)]) < 1e-5 and np.array_equal(t2, g["pre_tris"])


@pytest.mark.gpu
def test_device_alignment_large_vs_oracle():
    from animsnapbases_amd import HipEngine
    rng = np.random.default_rng(3)
    rest = rng.normal(size=(20000, 3))
    frames = []
    for f in range(12):
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        frames.append((rest + 0.01 * rng.normal(size=rest.shape)) @ q.T + rng.normal(size=3))
    frames = np.array(frames)
    e = HipEngine(0)
    al, T = e.align_frames(frames, True)
    e.close()
    al_o, T_o = orc.align_frames(frames, True)
    assert relerr(T, T_o) < 1e-10 and relerr(al.astype(np.float32), al_o) < 1e-6
    assert relerr(al, np.repeat(al[:1], 12, axis=0)) < 0.02       # frames really are aligned onto frame 0
