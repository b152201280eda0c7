"""pyani_amd — MI355X-native engine for pyani's hot path (TETRA now; ANIm next).

Mirrors the reference's module API for the path (pyani/tetra.py) on top of a C-ABI shared library
(include/pyani_gpu.h, built from pyani_amd/csrc/) that holds the hand-written gfx950 HIP kernels.
There is no CPU fallback: importing the compute modules without the built library raises.
"""
__version__ = "0.1.0"
