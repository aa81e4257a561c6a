"""Snapshot ingest -- mirror of the part of the reference's ``utils/process.py`` that feeds the hot
path: ``.off`` loading, sequence -> container, mesh pre-processing and the rigid
Procrustes alignment of every frame to frame 0.  The viewers (mayavi) are out of scope.

The alignment (``align``, process.py:235-250: per frame ``find_rbm_procrustes`` + ``transform``) runs
on the GPU (``asb_align_frames``, csrc/asb_ingest.hip): centroids, 3x3 cross-covariance, rotation
from the polar factor, transform -- one block per frame.  File parsing and the connected-component
clean-up are host work (text / graph code, no tensor arithmetic).

Containers: the reference writes ``.h5`` (h5py, gzip'd float32 ``verts`` + ``tris``); without h5py the
same keys go to ``.npz`` (``posSnapshots`` reads both).
"""
import os
import re
from glob import glob

import numpy as np


_NUM_RUN = re.compile(r"(\d+)")


def load_off(filename, no_colors=False):
    """Reads an ``OFF`` / ``COFF`` mesh (what process.py:46-67 returns: vertex array, [uint8 colours,] (M, 3) face array or
    None).  Parsed as ONE token stream -- comment lines and blank lines dropped, header, the three counts, then
    ``n_verts`` rows of equal width and ``n_faces`` rows ``k i0 .. i(k-1)`` -- instead of line slices fed to loadtxt."""
    with open(filename) as fh:
        rows = [ln.split() for ln in fh if ln.strip() and not ln.lstrip().startswith("#")]
    kind = rows[0][0] if rows and len(rows[0]) == 1 else None
    if kind not in ("OFF", "COFF"):
        raise AssertionError("OFF header missing")
    n_verts, n_faces = int(rows[1][0]), int(rows[1][1])
    body = rows[2:]
    table = np.array(body[:n_verts], dtype=np.float64).reshape(n_verts, -1)
    faces = None
    if n_faces > 0:
        faces = np.array(body[n_verts:n_verts + n_faces], dtype=np.int64)[:, 1:]
    if kind == "COFF":
        xyz, rgba = table[:, :3], table[:, 3:].astype(np.uint8)
    else:
        xyz, rgba = table, None
    return (xyz, faces) if no_colors else (xyz, rgba, faces)


def alphanum_key(s):
    """Natural-sort key (process.py:158-162): digit runs compare as numbers, ``"z23a" -> ["z", 23, "a"]``."""
    return [int(tok) if tok.isdigit() else tok for tok in _NUM_RUN.split(s)]


def sort_nicely(l):
    """In-place natural sort of file names (process.py:170-173)."""
    l.sort(key=alphanum_key)


def filter_reindex(condition, target):
    """New indices of ``target`` once only the elements with ``condition`` are kept (process.py:96-106)."""
    condition = np.asarray(condition)
    if condition.dtype != np.bool_:
        raise ValueError("condition must be a binary array")
    new_index = np.cumsum(condition, dtype=np.int64)
    new_index -= 1
    return new_index[target]


def _component_labels(n, edges):
    """Connected components of an undirected graph given as an (E, 2) edge array: label = the SMALLEST vertex index of the
    component (minimum-label propagation with pointer jumping, all NumPy; O(E log n)).  Ordering components by that
    label is the order in which a traversal from vertex 0 upwards meets them -- the numbering scipy's csgraph gives."""
    lab = np.arange(n, dtype=np.int64)
    a, b = edges[:, 0], edges[:, 1]
    while True:
        low = np.minimum(lab[a], lab[b])
        nxt = lab.copy()
        np.minimum.at(nxt, a, low)
        np.minimum.at(nxt, b, low)
        while True:                      # pointer jumping: every vertex adopts its label's label
            jump = nxt[nxt]
            if np.array_equal(jump, nxt):
                break
            nxt = jump
        if np.array_equal(nxt, lab):
            return lab
        lab = nxt


def preprocess_mesh_animation(verts, tris):
    """Mesh clean-up of process.py:107-148: triangles that are degenerate in frame 0 go, only the largest connected
    component stays (first one on a tie), the animation is centred and scaled into the unit cube (largest per-frame
    extent = 1).  Returns (verts, tris, removed_vertex_mask, centre, scale); arithmetic stays in ``verts``' dtype."""
    assert verts.ndim == 3 and tris.ndim == 2
    print("input mesh animation: %d frames, %d vertices, %d triangles" % (verts.shape[0], verts.shape[1], tris.shape[0]))
    n = verts.shape[1]
    p = verts[0]
    twice_area = np.linalg.norm(np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]]), axis=1)
    tris = tris[twice_area > 1.e-8]
    edges = np.concatenate([tris[:, (0, 1)], tris[:, (0, 2)], tris[:, (1, 2)]], axis=0)
    lab = _component_labels(n, edges)
    sizes = np.bincount(lab, minlength=n)
    n_comp = int(np.count_nonzero(sizes))
    if n_comp > 1:
        print("[warning] the mesh has %d connected components (sizes %s): only the largest is kept"
              % (n_comp, sizes[sizes > 0].tolist()))
    keep_vert = lab == int(np.argmax(sizes))         # first maximum = the tied component met first
    verts = verts[:, keep_vert, :]
    tris = filter_reindex(keep_vert, tris[keep_vert[tris].all(axis=1)])
    verts_mean = verts.mean(axis=0).mean(axis=0)
    verts = verts - verts_mean
    verts_scale = np.abs(verts.max(axis=1) - verts.min(axis=1)).max()
    verts = verts / verts_scale
    print("kept: %d vertices, %d triangles" % (verts.shape[1], tris.shape[0]))
    return verts, tris, ~keep_vert, verts_mean, verts_scale


def _write_container(path, verts, tris, **attrs):
    if os.path.splitext(path)[1].lower() == ".npz":
        np.savez_compressed(path, verts=verts, tris=tris, **attrs)
        return
    import h5py      # the reference's container (process.py:88-92): gzip'd 'verts', plain 'tris', attributes
    with h5py.File(path, 'w') as f:
        f.create_dataset('verts', data=verts, compression='gzip')
        f['tris'] = tris
        for k, v in attrs.items():
            f.attrs[k] = v


def _read_container(path):
    from .utils import read_animation
    return read_animation(path)


def convert_sequence_to_hdf5(filename_pattern, loader_function, hdf_output_file, max_frames, icreament):
    """Frame files -> one animation container (process.py:69-94; same signature, ``.npz`` accepted as well): every
    ``icreament``-th file of the naturally sorted match list, at most ``max_frames`` of them, one topology throughout,
    float32 vertices, pre-processed, stored with the centre and scale that were removed."""
    files = glob(os.path.expanduser(filename_pattern))
    sort_nicely(files)
    chosen = files[::icreament][:max_frames]
    frames, tris = [], None
    for f in chosen:
        print("reading %s" % f)
        v, t = loader_function(f)
        if tris is not None and t.shape != tris.shape:
            raise ValueError("inconsistent topology between meshes of different frames")
        frames.append(v)
        tris = t
    stack = np.asarray(frames, dtype=np.float32)
    stack, tris, _, centre, scale = preprocess_mesh_animation(stack, tris)
    _write_container(hdf_output_file, stack, tris, mean=centre, scale=scale)
    print("wrote %s" % hdf_output_file)


def transform(v, M, w=1):
    """Applies ``M`` to the points ``v`` (..., d) (process.py:196-208): a (d+1) x (d+1) matrix acts on the homogeneous
    points (v, w) and the result is de-homogenised; a d x d matrix is a plain linear map."""
    v = np.asarray(v)
    d = v.shape[-1]
    pts = v.reshape(-1, d)
    if M.shape == (d + 1, d + 1):
        lin = pts @ M[:d, :d].T + w * M[:d, d]
        hom = pts @ M[d, :d] + w * M[d, d]
        return (lin / hom[:, None]).reshape(v.shape)
    return (pts @ M.T).reshape(v.shape)


def align_frames(verts, rigid, engine=None):
    """The loop of ``align`` (process.py:241-246) on the GPU: returns (aligned float32 (F,N,3), T (F,4,4))."""
    from .engine import HipEngine
    eng = engine if engine is not None else HipEngine(0)
    out, T = eng.align_frames(np.asarray(verts, dtype=np.float64), rigid)
    return out.astype(np.float32), T


def find_rbm_procrustes(frompts, topts, rigid, engine=None):
    """process.py:210-234: the 4x4 rigid-body motion moving ``frompts`` onto ``topts`` (device)."""
    _, T = align_frames(np.stack([np.asarray(topts, dtype=np.float64), np.asarray(frompts, dtype=np.float64)]), rigid, engine)
    return T[1]


def align(input_hdf5_file, output_hdf5_file, rigid, engine=None):
    """Container in, aligned container out (process.py:235-250): every frame moved onto frame 0, all frames in one device call."""
    verts, tris = _read_container(input_hdf5_file)
    print("aligning %d frames onto frame 0 (%s)" % (len(verts), "rigid" if rigid else "affine"))
    out, _ = align_frames(verts, rigid, engine)
    _write_container(output_hdf5_file, out, tris)
