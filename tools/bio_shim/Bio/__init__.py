"""Minimal stand-in for Biopython, used ONLY by tools/make_goldens.py in the build
container to import the read-only Python reference (/root/reference/pyani/tetra.py).
It provides FASTA reading and reverse-complement; no arithmetic of the hot path lives here.
Never imported by the product (pyani_amd/) or by anything that runs on the GPU box."""
