"""Stand-in for the `intervaltree` package (absent from this image), just enough for pyani/anim.py:399-409 to run when
tools/make_goldens.py imports the reference: half-open intervals, from_tuples / merge_overlaps(strict=False) / iteration.
Holds no pyani code; never travels to the GPU box (tools/bio_shim is in .gpurunignore)."""
from collections import namedtuple

Interval = namedtuple("Interval", "begin end data")


class IntervalTree:
    def __init__(self, intervals=()):
        self._iv = set()
        for iv in intervals:
            if iv.begin >= iv.end:
                raise ValueError("IntervalTree: Null Interval objects not allowed in IntervalTree: {0}".format(iv))
            self._iv.add(iv)

    @classmethod
    def from_tuples(cls, tups):
        return cls(Interval(t[0], t[1], t[2] if len(t) > 2 else None) for t in tups)

    def merge_overlaps(self, data_reducer=None, data_initializer=None, strict=True):
        merged = []
        for iv in sorted(self._iv):
            if merged and (iv.begin < merged[-1].end or (not strict and iv.begin == merged[-1].end)):
                if iv.end > merged[-1].end:
                    merged[-1] = Interval(merged[-1].begin, iv.end, None)
            else:
                merged.append(Interval(iv.begin, iv.end, None))
        self._iv = set(merged)

    def __iter__(self):
        return iter(self._iv)

    def __len__(self):
        return len(self._iv)
