"""cutesdr_amd -- MI355X-native CuteSDR dsp/ receive chain.

The product is libcutesdr_mi.so (hand-written HIP kernels for gfx950 behind the C ABI of
include/cutesdr_mi.h) plus the header-only C++ drop-in classes in dropin/dsp/.  This Python
package is the host-side mirror of the reference's class surface used by tests and bench.py;
it is plumbing over the C ABI and contains no signal processing of its own.
"""
from . import _capi  # noqa: F401
from .host import *  # noqa: F401,F403

