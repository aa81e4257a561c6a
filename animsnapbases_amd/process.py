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
from io import StringIO

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import connected_components


def load_off(filename, no_colors=False):
    """process.py:46-67."""
    lines = open(filename).readlines()
    lines = [line for line in lines if line.strip() != '' and line[0] != '#']
    assert lines[0].strip() in ['OFF', 'COFF'], 'OFF header missing'
    has_colors = lines[0].strip() == 'COFF'
    n_verts, n_faces, _ = map(int, lines[1].split())
    vertex_data = np.loadtxt(StringIO(''.join(lines[2:2 + n_verts])), dtype=float)
    faces = np.loadtxt(StringIO(''.join(lines[2 + n_verts:])), dtype=int)[:, 1:] if n_faces > 0 else None
    colors = None
    if has_colors:
        colors = vertex_data[:, 3:].astype(np.uint8)
        vertex_data = vertex_data[:, :3]
    return (vertex_data, faces) if no_colors else (vertex_data, colors, faces)


def alphanum_key(s):
    """process.py:158-162: "z23a" -> ["z", 23, "a"]."""
    return [int(c) if c.isdigit() else c for c in re.split('([0-9]+)', s)]


def sort_nicely(l):
    """process.py:170-173."""
    l.sort(key=alphanum_key)


def filter_reindex(condition, target):
    """process.py:96-106."""
    if condition.dtype != bool:
        raise ValueError("condition must be a binary array")
    return (np.cumsum(condition) - 1)[target]


def preprocess_mesh_animation(verts, tris):
    """process.py:107-148: drop zero-area triangles (of frame 0), keep the biggest connected component,
    normalise the animation into the -0.5 .. 0.5 cube.  Returns (verts, tris, removed_mask, mean, scale)."""
    print("Vertices: ", verts.shape)
    print("Triangles: ", tris.shape)
    assert verts.ndim == 3 and tris.ndim == 2
    e1 = verts[0, tris[:, 1]] - verts[0, tris[:, 0]]
    e2 = verts[0, tris[:, 2]] - verts[0, tris[:, 0]]
    tris = tris[np.linalg.norm(np.cross(e1, e2), axis=1) > 1.e-8]
    ij = np.r_[np.c_[tris[:, 0], tris[:, 1]], np.c_[tris[:, 0], tris[:, 2]], np.c_[tris[:, 1], tris[:, 2]]]
    G = csr_matrix((np.ones(len(ij)), ij.T), shape=(verts.shape[1], verts.shape[1]))
    n_components, labels = connected_components(G, directed=False)
    if n_components > 1:
        size_components = np.bincount(labels)
        if len(size_components) > 1:
            print("[warning] found %d connected components in the mesh, keeping only the biggest one" % n_components)
            print("component sizes: ")
            print(size_components)
        keep_vert = labels == size_components.argmax()
    else:
        keep_vert = np.ones(verts.shape[1], bool)
    verts = verts[:, keep_vert, :]
    tris = filter_reindex(keep_vert, tris[keep_vert[tris].all(axis=1)])
    verts_mean = verts.mean(axis=0).mean(axis=0)
    verts -= verts_mean
    verts_scale = np.abs(np.ptp(verts, axis=1)).max()
    verts /= verts_scale
    print("after preprocessing:")
    print("Vertices: ", verts.shape)
    print("Triangles: ", tris.shape)
    return verts, tris, ~keep_vert, verts_mean, verts_scale


def _write_container(path, verts, tris, **attrs):
    if os.path.splitext(path)[1].lower() == ".npz":
        np.savez_compressed(path, verts=verts, tris=tris, **attrs)
        return
    import h5py      # the reference's container (process.py:88-92)
    with h5py.File(path, 'w') as f:
        f.create_dataset('verts', data=verts, compression='gzip')
        f['tris'] = tris
        for k, v in attrs.items():
            f.attrs[k] = v


def _read_container(path):
    from .utils import read_animation
    return read_animation(path)


def convert_sequence_to_hdf5(filename_pattern, loader_function, hdf_output_file, max_frames, icreament):
    """process.py:69-94 (output may also be ``.npz``)."""
    verts_all, tris = [], None
    files = glob(os.path.expanduser(filename_pattern))
    sort_nicely(files)
    count = 0
    for i, f in enumerate(files):
        if i % icreament == 0 and count < max_frames:
            print("loading file %d/%d [%s]" % (i + 1, len(files), f))
            verts, new_tris = loader_function(f)
            if tris is not None and new_tris.shape != tris.shape:
                raise ValueError("inconsistent topology between meshes of different frames")
            tris = new_tris
            verts_all.append(verts)
            count += 1
    verts_all = np.array(verts_all, np.float32)
    verts_all, tris, _, verts_mean, verts_scale = preprocess_mesh_animation(verts_all, tris)
    _write_container(hdf_output_file, verts_all, tris, mean=verts_mean, scale=verts_scale)
    print("saved as %s" % hdf_output_file)


def transform(v, M, w=1):
    """process.py:196-208."""
    v = np.asarray(v)
    if M.shape[0] == M.shape[1] == v.shape[-1] + 1:
        v1 = np.insert(v, v.shape[-1], w, axis=-1).reshape((-1, v.shape[-1] + 1))
        out = np.dot(v1, M.T)
        return (out[..., :-1] / out[..., np.newaxis, -1]).reshape(v.shape)
    return np.dot(v.reshape((-1, v.shape[-1])), M.T).reshape(v.shape)


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
    """process.py:235-250."""
    verts, tris = _read_container(input_hdf5_file)
    for i in range(len(verts)):
        print("frame %d/%d" % (i + 1, len(verts)))
    out, _ = align_frames(verts, rigid, engine)
    _write_container(output_hdf5_file, out, tris)
