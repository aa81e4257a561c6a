"""ctypes binding of ``libasb_hip.so`` (C ABI declared in ``include/asb.h``).

There is no CPU fallback: if the shared library is missing or no gfx950 GPU is usable,
``load()`` / ``HipContext()`` raise ``AsbLibraryError`` loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasb_hip.so")

c_dp = ctypes.c_void_p          # double* / generic pointers are passed as integers
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double

ASB_OK = 0
ERR_LIMIT = -4          # ASB_ERR_LIMIT
DEFLATE_RESIDUAL, DEFLATE_PROJECT = 0, 1

# name -> (restype, argtypes): must list every symbol of include/asb.h
PROTOTYPES = {
    "asb_abi_version": (c_int, []),
    "asb_create": (c_int, [c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "asb_destroy": (None, [ctypes.c_void_p]),
    "asb_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "asb_sync": (c_int, [ctypes.c_void_p]),
    "asb_prof_reset": (c_int, [ctypes.c_void_p, c_int]),
    "asb_prof_get": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl)]),
    "asb_snapshots_upload": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_i64, c_i64, c_i64, c_dp]),
    "asb_snapshots_adopt_dev": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_i64, c_dp, c_i64, c_i64]),
    "asb_snapshots_upload_rest": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_i64, c_i64, c_i64, c_dp, c_int, c_int, c_dp]),
    "asb_snapshots_adopt_dev_rest": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_i64, c_dp, c_i64, c_i64, c_int, c_int, c_dp]),
    "asb_snapshots_center": (c_int, [ctypes.c_void_p, c_int, c_int, ctypes.POINTER(c_dbl)]),
    "asb_snapshots_sqdev": (c_int, [ctypes.c_void_p, c_dbl, ctypes.POINTER(c_dbl)]),
    "asb_snapshots_scale": (c_int, [ctypes.c_void_p, c_dbl]),
    "asb_snapshots_get_mean": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_snapshots_download": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_deflate_begin": (c_int, [ctypes.c_void_p, c_i64, c_int, c_int]),
    "asb_deflate_xchg_len": (c_i64, [ctypes.c_void_p]),
    "asb_deflate_local_best": (c_int, [ctypes.c_void_p, c_i64, c_dp]),
    "asb_deflate_pick": (c_int, [ctypes.c_void_p, c_i64, c_dp, c_i64]),
    "asb_deflate_get_pick": (c_int, [ctypes.c_void_p, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl)]),
    "asb_deflate_apply": (c_int, [ctypes.c_void_p, c_i64, c_dp]),
    "asb_deflate_run_global": (c_int, [ctypes.c_void_p, c_i64, c_i64]),
    "asb_deflate_results": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "asb_st_upload": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "asb_st_residual_argmax": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl)]),
    "asb_deflate_residual_norm2": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_dbl)]),
    "asb_deim_block_residual_st": (c_int, [ctypes.c_void_p, c_i64, c_int, c_dp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_i64),
                                           ctypes.POINTER(c_dbl)]),
    "asb_deflate_reserve": (c_int, [ctypes.c_void_p, c_i64]),
    "asb_components_stream": (c_int, [ctypes.c_void_p, c_int]),
    "asb_components_stream_into": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64]),
    "asb_host_alloc": (c_int, [c_i64, ctypes.POINTER(ctypes.c_void_p)]),
    "asb_host_free": (c_int, [ctypes.c_void_p]),
    "asb_components_pinned": (c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "asb_panel_scale": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl), c_dbl]),
    "asb_panel_hist": (c_int, [ctypes.c_void_p, c_int, c_dp]),
    "asb_panel_tau": (c_int, [ctypes.c_void_p, c_int, c_dp]),
    "asb_panel_top_energies": (c_int, [ctypes.c_void_p, c_dp, c_i64]),
    "asb_panel_global_tau": (c_int, [ctypes.c_void_p, c_dp, c_int, c_i64, c_dp]),
    "asb_panel_set_tau": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_panel_target": (c_i64, [ctypes.c_void_p]),
    "asb_panel_select": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_int, c_dp, c_dp, ctypes.POINTER(c_i64),
                                 ctypes.POINTER(c_int)]),
    "asb_panel_capacity": (c_i64, [ctypes.c_void_p]),
    "asb_panel_assemble": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_int, c_i64]),
    "asb_panel_assemble_packed": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_int, c_i64]),
    "asb_panel_run": (c_int, [ctypes.c_void_p, c_i64, c_int, c_int, c_int, ctypes.POINTER(c_i64)]),
    "asb_panel_project": (c_int, [ctypes.c_void_p, c_i64, c_int]),
    "asb_panel_run_spec": (c_int, [ctypes.c_void_p, c_i64, c_int, c_int, c_int, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "asb_panel_project_spec": (c_int, [ctypes.c_void_p, c_i64, c_int, c_int, ctypes.POINTER(c_i64)]),
    "asb_panel_commit": (c_int, [ctypes.c_void_p, c_i64, c_int]),
    "asb_panel_project_spec_dev": (c_int, [ctypes.c_void_p, c_i64, c_int, c_int, ctypes.c_void_p]),
    "asb_panel_refresh": (c_int, [ctypes.c_void_p, c_i64, ctypes.POINTER(c_dbl), ctypes.POINTER(c_i64)]),
    "asb_deflate_stats": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "asb_deflate_spec_stats": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "asb_energy_block_argmax": (c_int, [ctypes.c_void_p, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl)]),
    "asb_components_expand": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_i64, c_dp]),
    "asb_pod_slices": (c_int, [ctypes.c_void_p, c_int, c_i64]),
    "asb_geodesic_coarse_setup": (c_int, [ctypes.c_void_p, c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dbl]),
    "asb_deim_run": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64), c_dp, ctypes.POINTER(c_int)]),
    "asb_deim_block_residual": (c_int, [ctypes.c_void_p, c_i64, c_int, c_dp, ctypes.POINTER(c_dbl)]),
    "asb_sym_eig_topk": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp, ctypes.POINTER(c_i64)]),
    "asb_pod_basis_dev": (c_int, [ctypes.c_void_p, c_i64]),
    "asb_pod_rotate": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_dp]),
    "asb_pod_power": (c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "asb_pod_deflate_begin": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64]),
    "asb_pod_deflate_end": (c_int, [ctypes.c_void_p, c_i64]),
    "asb_qr_apply_joint": (c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "asb_test_tridiag_eig": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_i64, c_i64, c_dp, c_dp, ctypes.POINTER(c_i64)]),
    "asb_test_jacobi_rows": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_i64, c_dp, c_dp, ctypes.POINTER(c_i64)]),
    "asb_test_chol_tinv": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_dp]),
    "asb_deflate_energy_passes": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64)]),
    "asb_fetch_double": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(c_dbl)]),
    "asb_panel_set_coop": (c_int, [ctypes.c_void_p, c_int]),
    "asb_deflate_coop_fallbacks": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64)]),
    "asb_deflate_guessed_panels": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64)]),
    "asb_deflate_sketch_stats": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "asb_project_switch_residual": (c_int, [ctypes.c_void_p, c_i64]),
    "asb_splocs_trace_begin": (c_int, [ctypes.c_void_p, c_i64]),
    "asb_splocs_objective_dev": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_i64]),
    "asb_splocs_trace": (c_int, [ctypes.c_void_p, c_i64, c_dp]),
    "asb_panel_read_run": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_int, c_int, ctypes.POINTER(c_int), ctypes.c_void_p,
                                   ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "asb_panel_read_commit": (c_int, [ctypes.c_void_p, c_dp, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "asb_fetch_doubles": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int, c_dp]),
    "asb_deflate_switch_stats": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_i64)]),
    "asb_panel_guess_stats": (c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(c_int)]),
    "asb_panel_guess_begin": (c_int, [ctypes.c_void_p, c_int]),
    "asb_panel_guess_end": (c_int, [ctypes.c_void_p]),
    "asb_panel_sub_run": (c_int, [ctypes.c_void_p, c_int, c_i64, c_int, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                  ctypes.POINTER(c_int)]),
    "asb_panel_sub_project": (c_int, [ctypes.c_void_p, c_i64, c_int, ctypes.POINTER(c_int)]),
    "asb_panel_sub_check": (c_int, [ctypes.c_void_p, c_int, c_i64, c_int, ctypes.c_void_p]),
    "asb_panel_sub_commit": (c_int, [ctypes.c_void_p, c_int, c_i64, c_int, c_int]),
    "asb_test_l2w_probe": (c_int, [ctypes.c_void_p, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_double)]),
    "asb_deflate_download_residual": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_deflate_block_argmax": (c_int, [ctypes.c_void_p, c_int, c_dp, c_dp]),
    "asb_deflate_force_next": (c_int, [ctypes.c_void_p, c_i64]),
    "asb_components_post": (c_int, [ctypes.c_void_p, c_int, c_dbl, c_dp, c_dp]),
    "asb_orth_gram": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_orth_apply": (c_int, [ctypes.c_void_p, c_dp, c_dp]),
    "asb_orth_refine": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_orth_gram_get": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_components_transform": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_components_download": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_components_upload": (c_int, [ctypes.c_void_p, c_dp, c_i64]),
    "asb_geodesic_setup": (c_int, [ctypes.c_void_p, c_int, c_int] + [c_dp] * 14),
    "asb_geodesic_solve": (c_int, [ctypes.c_void_p, c_dp, c_int, c_dbl, c_dp, c_dp]),
    "asb_geodesic_dense_setup": (c_int, [ctypes.c_void_p]),
    "asb_geodesic_bt_setup": (c_int, [ctypes.c_void_p, c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "asb_deflate_apply_geodesic": (c_int, [ctypes.c_void_p, c_i64, c_dbl, c_dbl]),
    "asb_align_frames": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_i64, c_int, c_dp]),
    "asb_pod_gram": (c_int, [ctypes.c_void_p, c_dp, c_dp]),
    "asb_pod_basis": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_i64]),
    "asb_pod_project": (c_int, [ctypes.c_void_p, c_dp, c_dp]),
    "asb_components_truncate": (c_int, [ctypes.c_void_p, c_i64]),
    "asb_sym_tridiag": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_dp, c_dp]),
    "asb_sym_backtransform": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "asb_snapshots_affine": (c_int, [ctypes.c_void_p, c_dbl, c_int, c_dp]),
    "asb_qr_apply": (c_int, [ctypes.c_void_p, c_dp]),
    "asb_deim_step": (c_int, [ctypes.c_void_p, c_i64, c_dp, ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl)]),
    "asb_deim_row": (c_int, [ctypes.c_void_p, c_i64, c_dp]),
    "asb_splocs_begin": (c_int, [ctypes.c_void_p]),
    "asb_splocs_gram": (c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.POINTER(c_dbl)]),
    "asb_splocs_weights": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_dp]),
    "asb_splocs_admm": (c_int, [ctypes.c_void_p, c_dp, c_dbl, c_int]),
    "asb_splocs_admm_fields": (c_int, [ctypes.c_void_p, c_dp, c_dbl, c_dbl, c_dbl, c_dbl, c_int]),
    "asb_geodesic_cache_add": (c_int, [ctypes.c_void_p, c_dp, c_int, c_dbl, c_dp]),
    "asb_geodesic_cache_clear": (c_int, [ctypes.c_void_p]),
    "asb_splocs_objective": (c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl),
                                     ctypes.POINTER(c_dbl)]),
    "asb_splocs_results": (c_int, [ctypes.c_void_p, c_dp, c_dp]),
    "asb_test_eig3": (None, [c_dp, c_dp]),
    "asb_test_sketch_predict": (c_int, [ctypes.c_void_p, c_dp, c_dp, c_dp, c_i64, c_int, c_int, c_dp, ctypes.POINTER(c_i64),
                                        ctypes.POINTER(c_int)]),
    "asb_test_spd_inverse": (c_int, [ctypes.c_void_p, c_dp, c_i64, c_dp]),
}


class AsbLibraryError(RuntimeError):
    pass


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so.  A process must end up with ONE HIP runtime: if this library pulled in
    the system one first, torch (the multi-GPU plumbing) would later load a second copy and find no GPU.  So when torch
    is installed its runtime is loaded first -- without importing torch -- and libasb_hip.so binds to it by SONAME."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                              # torch already loaded its runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load():
    """Loads the HIP library and binds every entry point of include/asb.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsbLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C animsnapbases_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    _preload_torch_hip_runtime()
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise AsbLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise AsbLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    """Host pointer of a C-contiguous float64/int64 array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
