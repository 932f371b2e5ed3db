"""mba-vo_amd: MI355X-native blur-aware photometric tracking path (hot path of MBA-VO's ba_tracker).

The product is the HIP/C++ shared library `libmbavo.so` built from `csrc/` (see
include/mbavo.h for its C ABI and csrc/ba_tracker.h for the reference-compatible
C++ API).  This Python package is only plumbing for tests and bench.py:
`capi` binds the C ABI with ctypes, `synth` builds deterministic synthetic inputs.
The directory name has a hyphen, so import it through the repo-root shim
`mba_vo_amd` (or importlib).
"""
from . import synth  # noqa: F401
from . import capi  # noqa: F401
from .capi import load, build, LIB_PATH  # noqa: F401
