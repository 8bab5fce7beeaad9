"""slimfastq_amd -- MI355X-native implementation of slimfastq's entropy-coding hot path.

The product is the C-ABI shared library (include/slimfastq_amd.h, built by slimfastq_amd.build from
csrc/) and the C++ CLI next to it.  This Python package is a thin ctypes binding used by the tests
and bench.py; it contains no coding logic and no CPU fallback: importing `capi` without the built
library, or creating a context without a HIP device, raises.
"""
__all__ = ["capi", "build"]
