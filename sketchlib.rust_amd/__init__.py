"""sketchlib.rust_amd -- MI355X (gfx950) implementation of sketchlib.rust's pairwise
distance path (src/distances/* of the reference).

The product is the C-ABI shared library built from csrc/ (include/sketchlib_dist.h).
This Python package is plumbing around it: the build recipe, a ctypes binding used by
the tests and bench.py, a synthetic sketch generator, and the multi-GPU row partition
over torch.distributed.  Nothing here computes distances on the CPU.
"""
from . import build as _build_mod  # noqa: F401
from .build import ab_library_path, build_ab_library, build_clock_probe, build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path", "build_ab_library", "ab_library_path", "build_clock_probe"]
