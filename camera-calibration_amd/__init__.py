"""MI355X-native calibration LM core: host mirror of the reference's call surface.

The compute lives in csrc/ (hand-written HIP for gfx950 behind a C ABI, see
include/ccalib.h); this package is the thin Python host side used by tests and
bench.py.  It never falls back to a CPU implementation.
"""
from . import synth  # noqa: F401
