"""animsnapbases_amd -- MI355X-native implementation of the animSnapBases snapshot-reduction
hot path (posSnapshots / posComponents), as a drop-in for that path.

Python host code (this package) -> ctypes -> ``libasb_hip.so`` (hand-written HIP for gfx950).
Importing the package does not touch the GPU; constructing ``posSnapshots`` does, and
fails loudly if the HIP library or a gfx950 device is missing (no CPU fallback).
"""
from ._lib import AsbLibraryError, LIB_PATH, load as load_library      # noqa: F401
from .constraints import constraintsComponents, nonlinearSnapshots      # noqa: F401
from .distributed import Comm, partition                                # noqa: F401
from .engine import HipEngine                                           # noqa: F401
from .geodesic import GeodesicDistanceComputation                       # noqa: F401
from .posComponents import posComponents                                # noqa: F401
from .posSnapshots import posSnapshots                                  # noqa: F401

__all__ = ["posSnapshots", "posComponents", "nonlinearSnapshots", "constraintsComponents", "GeodesicDistanceComputation", "HipEngine", "Comm", "partition",
           "load_library", "AsbLibraryError", "LIB_PATH"]
