"""pyani_amd — MI355X-native engine for pyani's hot path (TETRA, ANIm, ANIb fragment mode).

Mirrors the reference's module API for the path (pyani/tetra.py) on top of a C-ABI shared library
(include/pyani_gpu.h, built from pyani_amd/csrc/) that holds the hand-written gfx950 HIP kernels.
There is no CPU fallback: importing the compute modules without the built library raises.
"""
__version__ = "0.1.0"


def configure_runtime(hw_queues: int = 8) -> bool:
    """See pyani_amd._lib.configure_runtime: GPU_MAX_HW_QUEUES for jobs that also run RCCL (explicit, never at import)."""
    from . import _lib
    return _lib.configure_runtime(hw_queues)
