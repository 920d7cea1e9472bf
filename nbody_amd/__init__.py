"""MI355X-native gravity engine for the quillaja/nbody hot path: force evaluation (tiled direct sum and
Barnes-Hut build + walk) with the integrator update fused into the force kernel, behind a C ABI
(include/nbody_engine.h).  This package is the Python host side: a ctypes binding of the shared library,
a mirror of the reference's body/physics interface, seeded initial conditions and the multi-GPU stepper."""
from .engine import Engine, EngineError, Stats, library_path, load_library  # noqa: F401
from . import physics  # noqa: F401

__all__ = ["Engine", "EngineError", "Stats", "library_path", "load_library", "physics"]
