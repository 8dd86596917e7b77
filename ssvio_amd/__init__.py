"""ssvio_amd -- MI355X (gfx950) compute core for the stereo-SLAM hot path of weihaoysgs/ssvio.

The product is libssx.so (HIP kernels behind the C ABI of include/ssx.h); this package is the thin
Python host layer used by tests and bench.py.  Nothing here computes on the CPU.
"""
from ._lib import Context, SsxError, load  # noqa: F401

__all__ = ["Context", "SsxError", "load"]
