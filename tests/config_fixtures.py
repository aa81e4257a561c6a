"""Helpers shared by the CPU and GPU tests of the full-size config fixtures (``tests/golden/c2_*.npz``, ``c3_*.npz``,
``c4_*.npz``, ``c5_*.npz``; written by oracle/gen_golden_configs.py from the unmodified reference)."""
import types

import numpy as np

from conftest import relerr
from oracle import asb_oracle as orc

NPROJ, NSAMPLE = 32, 48


def regen_frames(g):
    """The (F, N, 3) frames the reference ran on: rebuilt from the seed on the fixture's rest mesh."""
    return orc.synth_snapshots(g["rest"], int(g["F"]), rank=int(g["rank"]), noise=float(g["noise"]),
                               seed=int(g["seed"]), kind=str(g["kind"]))


def probes(g):
    """Same seeded probe matrices / vertex sample as the generator (gen_golden_configs.probes)."""
    N, F = (g["rest"].shape[0] if "rest" in g else int(g["N"])), int(g["F"])
    rng = np.random.default_rng(int(g["probe_seed"]))
    G = rng.normal(size=(3 * N, NPROJ)) / np.sqrt(3 * N)
    H = rng.normal(size=(NPROJ, F)) / np.sqrt(F)
    sv = np.sort(rng.choice(N, size=NSAMPLE, replace=False))
    assert np.array_equal(sv, g["sample_verts"])
    return G, H, sv


def make_param(g, **over):
    d = {}
    for k, v in g.items():
        if k.startswith("param_"):
            v = v.item() if v.ndim == 0 else v
            d[k[6:]] = str(v) if isinstance(v, (np.str_, bytes, str)) else v
    d.update(over)
    return types.SimpleNamespace(**d)


def check_deflation(g, psf, mean, idx, comps, weigs, measures, signed, tol, mtol, csv=None):
    """``signed``: local support fixes each component's sign; global support leaves LAPACK's (aligned here through
    the projections)."""
    G, H, sv = probes(g)
    K = comps.shape[0]
    assert abs(psf - float(g["pre_scale_factor"])) < 1e-12 * float(g["pre_scale_factor"])
    assert relerr(mean.reshape(-1) @ G, g["mean_proj"]) < 1e-12
    assert np.asarray(idx).tolist() == g["idx"][:K].tolist()                    # bit-exact index selection
    P = comps.reshape(K, -1) @ G
    sign = np.ones(K) if signed else np.sign(np.einsum("kp,kp->k", P, g["comps_proj"][:K]))
    assert relerr(P * sign[:, None], g["comps_proj"][:K]) < tol
    assert relerr(comps[:, sv, :] * sign[:, None, None], g["comps_sample"][:K]) < tol
    assert relerr(np.sqrt((comps.reshape(K, -1) ** 2).sum(1)), g["comps_norms"][:K]) < tol
    assert relerr((H @ weigs) * sign[None, :], g["weigs_proj"][:, :K]) < tol
    assert relerr(weigs[:8] * sign[None, :], g["weigs_head"][:, :K]) < tol
    assert relerr(np.sqrt((weigs ** 2).sum(0)), g["weigs_norms"][:K]) < tol
    assert relerr(measures, g["measures"][:K]) < mtol
    if csv is not None:
        lines = csv.splitlines()
        ref_lines = str(g["csv_text"]).splitlines()
        assert lines[0] == ref_lines[0] == "component,singVal,norm_R"
        rows = np.array([[float(x) for x in ln.split(",")] for ln in lines[1:] if ln])
        ref_rows = np.array([[float(x) for x in ln.split(",")] for ln in ref_lines[1:] if ln])
        assert rows.shape == ref_rows[:K].shape and relerr(rows, ref_rows[:K]) < mtol


def c4_frames(g):
    """Config 4's input, rebuilt on the host from the seed (oracle.synth_uniform_snapshots), checked against the corner
    values the generator recorded."""
    verts, _ = orc.synth_uniform_snapshots(int(g["F"]), int(g["N"]), int(g["seed"]))
    assert np.array_equal(verts[0, :4], g["frame0_head"]) and np.array_equal(verts[-1, -4:], g["frame_last_tail"])
    return verts


def c5_frames(g):
    frames = orc.synth_constraint_frames(int(g["F"]), int(g["ep"]), int(g["rank"]), float(g["decay"]), float(g["noise"]),
                                         int(g["seed"]))
    assert np.array_equal(frames[0, :4], g["frame0_head"]) and np.array_equal(frames[-1, -4:], g["frame_last_tail"])
    return frames


def c5_probes(g):
    M = 3 * int(g["ep"])
    rng = np.random.default_rng(int(g["probe_seed"]))
    Gp = rng.normal(size=(M, NPROJ)) / np.sqrt(M)
    sv = np.sort(rng.choice(int(g["ep"]), size=NSAMPLE, replace=False))
    assert np.array_equal(sv, g["sample_rows"])
    return Gp, sv
