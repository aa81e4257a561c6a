"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Makes the *unmodified* reference (``/root/reference``, pure Python) importable in the
build container so that ``oracle/gen_golden.py`` can run it and record golden vectors.

The reference pulls GUI / mesh libraries at import time that this image lacks
(h5py, igl, trimesh, polyscope, mayavi, traits, tvtk, pyface ...), and
``utils/utils.py`` runs a plotting demo at import.  We pre-seed
``sys.modules`` with inert stand-ins for those *third-party GUI/mesh packages only*;
no reference file is copied, patched or shadowed, and nothing on the numeric path
(numpy / scipy) is replaced.

``/root/reference`` does not exist on the GPU box: this module is only ever used here,
to produce ``tests/golden/*.npz``.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("ASB_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "snapbases"))


class _Dummy(object):
    """Do-nothing object: callable, and every attribute is another _Dummy."""

    def __call__(self, *a, **k):
        return None

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _Inert(types.ModuleType):
    """Module whose every attribute is a do-nothing callable / class factory."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


def _stub(name, **attrs):
    m = _Inert(name)
    m.__path__ = []  # behave as a package so that sub-module imports resolve
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _raiser(msg):
    def f(*a, **k):
        raise RuntimeError(msg)

    return f


def install_stubs():
    if "h5py" not in sys.modules:
        try:
            import h5py  # noqa: F401
        except Exception:
            _stub("h5py", File=_raiser("h5py is not installed in this image"))
    _stub("trimesh", load=_raiser("trimesh stub: no mesh loading in the oracle harness"))
    _stub("igl", MASSMATRIX_TYPE_VORONOI=1, massmatrix=_raiser("igl stub"))
    _stub("polyscope")
    _stub("potpourri3d")
    _stub("pygame")
    for n in ("mayavi", "mayavi.mlab", "mayavi.tools", "mayavi.tools.mlab_scene_model",
              "mayavi.core", "mayavi.core.ui", "mayavi.core.ui.mayavi_scene",
              "mayavi.core.api"):
        _stub(n)

    class HasTraits(object):
        pass

    def _decorator_factory(*a, **k):
        def deco(fn):
            return fn

        return deco

    _stub("traits")
    _stub("traits.api", HasTraits=HasTraits, on_trait_change=_decorator_factory)
    _stub("traitsui")
    _stub("traitsui.api")
    for n in ("tvtk", "tvtk.api", "tvtk.common", "tvtk.pyface", "tvtk.pyface.scene_editor",
              "pyface", "pyface.timer", "pyface.timer.api", "pyface.api"):
        _stub(n)


def import_reference():
    """Returns a dict with the reference classes / functions on the hot path."""
    assert reference_available(), REF_ROOT
    os.environ.setdefault("MPLBACKEND", "Agg")
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # utils/utils.py executes a demo at its last line; everything above stays bound
    # when that demo fails on the stubbed trimesh.load.
    if "utils.utils" not in sys.modules or not hasattr(sys.modules["utils.utils"], "store_components"):
        import utils  # the reference's namespace package

        spec = importlib.util.spec_from_file_location("utils.utils", os.path.join(REF_ROOT, "utils", "utils.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["utils.utils"] = mod
        utils.utils = mod
        try:
            spec.loader.exec_module(mod)
        except RuntimeError as e:  # the import-time demo
            if "stub" not in str(e):
                raise
    from snapbases.posComponents import posComponents
    from snapbases.posSnapshots import posSnapshots
    from utils.support import GeodesicDistanceComputation
    from snapbases.constraintsComponents import constraintsComponents
    from snapbases.nonlinear_snapshots import nonlinearSnapshots
    import utils.utils as uu

    return dict(posComponents=posComponents, posSnapshots=posSnapshots,
                GeodesicDistanceComputation=GeodesicDistanceComputation,
                constraintsComponents=constraintsComponents,
                nonlinearSnapshots=nonlinearSnapshots,
                store_components=uu.store_components, read_obj=uu.read_obj,
                utils=uu)
