"""The object model of MUMmer's .delta / .filter files that pyani keeps for its nucmer output (pyani/nucmer.py:47-351: DeltaData,
DeltaMetadata, DeltaHeader, DeltaAlignment, DeltaComparison, DeltaIterator) — same class names, attributes, equality rules and
`str()` rendering, so that code written against the reference's module (tests/tools.py assertNucmerEqual, report code that walks
comparisons and alignments) runs unchanged on the files this package writes (pyani_amd.anim.write_delta) or reads.

Equality follows the reference: two DeltaData are equal when the program and the comparisons are (file paths on line 1 are not
compared); two DeltaComparison when their headers are equal and their alignments are equal AS SORTED LISTS (MUMmer builds
differ in output order); two DeltaAlignment when their four coordinates are equal.  `DeltaAlignment.identical()` /
`DeltaData.identical()` are the stricter checks this package's tests use: error counts and indel lists too.

File grammar: line 1 = the two FASTA paths, line 2 = the program name, then per sequence pair a header
`>ref qry reflen qrylen` followed by alignments `rs re qs qe errs simerrs stops` each with its indel offsets, one per line, and a
terminating `0`.
"""
import gzip
import os
from pathlib import Path
from typing import Iterator, List, Optional, TextIO, Union


class DeltaMetadata:
    """Line 1 (reference and query FASTA paths) and line 2 (program) of a .delta file."""

    def __init__(self) -> None:
        self.reference: Optional[Path] = None
        self.query: Optional[Path] = None
        self.program: Optional[str] = None

    def __eq__(self, other):
        return isinstance(other, DeltaMetadata) and (self.reference, self.query, self.program) == (other.reference, other.query, other.program)

    def __str__(self):
        return os.linesep.join([f"{self.reference} {self.query}", str(self.program)])


class DeltaHeader:
    """`>ref qry reflen qrylen`: the two sequences of a comparison."""

    def __init__(self, reference: str, query: str, reflen: int, querylen: int) -> None:
        self.reference = reference[1:] if reference.startswith(">") else reference
        self.query = query
        self.referencelen = int(reflen)
        self.querylen = int(querylen)

    def _key(self):
        return (self.reference, self.query, self.referencelen, self.querylen)

    def __eq__(self, other):
        return isinstance(other, DeltaHeader) and self._key() == other._key()

    def __str__(self):
        return f">{self.reference} {self.query} {self.referencelen} {self.querylen}"


class DeltaAlignment:
    """One alignment: coordinates (1-based, inclusive; reverse-strand hits have querystart > queryend), error counts and the
    indel offsets (the reference keeps them as the file's strings; so does this class, `indel_offsets` gives them as ints)."""

    def __init__(self, refstart, refend, qrystart, qryend, errs, simerrs, stops) -> None:
        self.refstart, self.refend = int(refstart), int(refend)
        self.querystart, self.queryend = int(qrystart), int(qryend)
        self.errs, self.simerrs, self.stops = int(errs), int(simerrs), int(stops)
        self.indels: List[str] = []

    def _coords(self):
        return (self.refstart, self.refend, self.querystart, self.queryend)

    def __lt__(self, other):
        return self._coords() < other._coords()

    def __eq__(self, other):
        return isinstance(other, DeltaAlignment) and self._coords() == other._coords()

    __hash__ = None

    @property
    def indel_offsets(self) -> List[int]:
        """The signed offsets without the terminating 0."""
        vals = [int(x) for x in self.indels]
        return vals[:-1] if vals and vals[-1] == 0 else vals

    def identical(self, other) -> bool:
        """Coordinates, error counts AND the indel list equal (the reference's == compares coordinates only)."""
        return (self == other and (self.errs, self.simerrs, self.stops) == (other.errs, other.simerrs, other.stops)
                and self.indel_offsets == other.indel_offsets)

    def __str__(self):
        head = f"{self.refstart} {self.refend} {self.querystart} {self.queryend} {self.errs} {self.simerrs} {self.stops}"
        return os.linesep.join([head] + [str(x) for x in self.indels])


class DeltaComparison:
    """A header and its alignments."""

    def __init__(self, header: DeltaHeader, alignments: List[DeltaAlignment]) -> None:
        self.header = header
        self.alignments = alignments

    def add_alignment(self, aln: DeltaAlignment) -> None:
        self.alignments.append(aln)

    def __eq__(self, other):
        return isinstance(other, DeltaComparison) and self.header == other.header and sorted(self.alignments) == sorted(other.alignments)

    def identical(self, other) -> bool:
        if not (isinstance(other, DeltaComparison) and self.header == other.header and len(self) == len(other)):
            return False
        key = lambda a: a._coords() + (a.errs,)      # noqa: E731
        return all(a.identical(b) for a, b in zip(sorted(self.alignments, key=key), sorted(other.alignments, key=key)))

    def __len__(self):
        return len(self.alignments)

    def __str__(self):
        return os.linesep.join([str(self.header)] + [str(a) for a in self.alignments])


class DeltaIterator:
    """Iterates a .delta / .filter handle: first the DeltaMetadata, then one DeltaComparison per `>` block."""

    def __init__(self, handle: TextIO) -> None:
        self._elements = self._parse(handle)

    @staticmethod
    def _parse(handle: TextIO) -> Iterator[Union[DeltaMetadata, DeltaComparison]]:
        meta = DeltaMetadata()
        first = handle.readline().split()
        if len(first) >= 2:
            meta.reference, meta.query = Path(first[0]), Path(first[1])
        meta.program = handle.readline().strip()
        yield meta
        comparison: Optional[DeltaComparison] = None
        alignment: Optional[DeltaAlignment] = None
        for line in handle:
            f = line.split()
            if not f:
                continue
            if f[0].startswith(">"):
                if comparison is not None:
                    yield comparison
                comparison = DeltaComparison(DeltaHeader(*f[:4]), [])
                alignment = None
            elif len(f) == 1:
                if alignment is None:
                    raise ValueError(f"delta file: indel line {line.strip()!r} before any alignment header")
                alignment.indels.append(f[0])
            else:
                # (the reference unpacks the 7 fields and raises on anything else, pyani/nucmer.py DeltaIterator)
                if len(f) != 7 or comparison is None:
                    raise ValueError(f"delta file: malformed alignment line {line.strip()!r}")
                alignment = DeltaAlignment(*f)
                comparison.add_alignment(alignment)
        if comparison is not None:
            yield comparison

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._elements)


class DeltaData:
    """A whole .delta / .filter file."""

    def __init__(self, name: str, handle: Optional[TextIO] = None) -> None:
        self.name = name
        self._metadata: Optional[DeltaMetadata] = None
        self._comparisons: List[DeltaComparison] = []
        if handle is not None:
            self.from_delta(handle)

    @classmethod
    def from_file(cls, path) -> "DeltaData":
        opener = gzip.open if str(path).endswith(".gz") else open
        with opener(path, "rt") as fh:
            return cls(Path(path).name, fh)

    def from_delta(self, handle: TextIO) -> None:
        for element in DeltaIterator(handle):
            if isinstance(element, DeltaMetadata):
                self._metadata = element
            else:
                self._comparisons.append(element)

    comparisons = property(lambda self: self._comparisons)
    metadata = property(lambda self: self._metadata)
    reference = property(lambda self: self._metadata.reference)
    query = property(lambda self: self._metadata.query)
    program = property(lambda self: self._metadata.program)

    def __eq__(self, other):
        return isinstance(other, DeltaData) and self.program == other.program and self._comparisons == other._comparisons

    def identical(self, other, ordered: bool = False) -> bool:
        """Every comparison with every alignment's error counts and indel list; ordered=False: the comparisons as sets by header."""
        if not isinstance(other, DeltaData) or self.program != other.program or len(self) != len(other):
            return False
        mine, theirs = (self._comparisons, other._comparisons) if ordered else (
            sorted(self._comparisons, key=lambda c: c.header._key()), sorted(other._comparisons, key=lambda c: c.header._key()))
        return all(a.identical(b) for a, b in zip(mine, theirs))

    def __len__(self):
        return len(self._comparisons)

    def __str__(self):
        return os.linesep.join([str(self._metadata)] + [str(c) for c in self._comparisons])
