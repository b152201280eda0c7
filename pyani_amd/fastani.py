"""pyani_amd.fastani — the SKETCH mode behind the interface of pyani/fastani.py.

pyani's fastANI support shells out to the third-party `fastANI` program once per ordered pair (fastani.py:139-229:
generate_fastani_commands / construct_fastani_cmdline, `--fragLen 3000 -k 16 --minFraction 0.2`) and parses the one-line result file
(fastani.py:231-270: parse_fastani_file -> ComparisonResult(reference, query, ani, matches, fragments)).  Here the estimate is computed
in-process on the GPU from the packed genomes already resident in HBM (pg_sketch_pairs, pyani_amd/csrc/pg_sketch.hip), with the same
parameters, the same result tuple, the same result-file line and the same failure for pairs without a result — but by an estimator
of its own (FracMinHash containment per query fragment: pyani_amd/csrc/pg_sketch_core.h), NOT by fastANI's MashMap pipeline: the
numbers are fastANI-SHAPED estimates with their own error bar, kept in their own columns and files and never
written into the exact ANIm / ANIb matrices.  No CPU fallback: the functions that compute need an Engine."""
from pathlib import Path
from typing import Dict, Iterable, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np


class PyaniFastANIException(Exception):
    """Exception raised when there is a problem with a sketch-mode result (the reference's class of the same name)."""


class ComparisonResult(NamedTuple):
    """The reference's result tuple (fastani.py:59-66): file names, ANI as a FRACTION, matching fragments, query fragments."""

    reference: Path
    query: Path
    ani: float
    matches: int
    fragments: int


def get_version() -> str:
    """What a Comparison row records as program version: this engine's sketch mode, not a fastANI binary."""
    from . import _lib
    return f"pyani_amd sketch mode (FracMinHash containment, k = 16) / {_lib.load().pg_version().decode()}"


def calculate_fastani_pairs(engine, qry_ids: Sequence[int], ref_ids: Sequence[int], fragLen: int = 3000, kmerSize: int = 16,
                            minFraction: float = 0.2, scale: int = 16) -> np.ndarray:
    """The estimates of many ordered pairs in one call (one record per pair: ani, matches, fragments, status)."""
    if kmerSize != 16:
        raise PyaniFastANIException("the sketch mode works on 16-mers (fastANI's default -k 16) only")
    return engine.sketch_pairs(qry_ids, ref_ids, frag_len=fragLen, scale=scale, min_fraction=minFraction)


def comparison_results(engine, files: Sequence[Path], ids: Sequence[int], fragLen: int = 3000, kmerSize: int = 16,
                       minFraction: float = 0.2) -> Dict[Tuple[str, str], Optional[ComparisonResult]]:
    """What pyani's loop over generate_fastani_commands + parse_fastani_file yields for an input set: every ordered pair INCLUDING a
    genome against itself (fastani.py:166-184 runs query x reference over the whole file list), keyed (query stem, reference stem);
    None where fastANI would have written an empty file."""
    files = [Path(f) for f in files]
    q = [ids[i] for i in range(len(files)) for _ in files]
    r = [ids[j] for _ in files for j in range(len(files))]
    res = calculate_fastani_pairs(engine, q, r, fragLen, kmerSize, minFraction)
    out, k = {}, 0
    for fq in files:
        for fr in files:
            x = res[k]
            # field order = the reference's positional quirk (parse_fastani_file passes the file's columns through: the QUERY file
            # lands in `.reference`, fastani.py:262-270; subcmd_fastani.py:447 unpacks it as `query, ref, ...`): results taken
            # in-process equal results read back from write_fastani_file
            out[(fq.stem, fr.stem)] = None if int(x["status"]) else ComparisonResult(fq, fr, float(x["ani"]), int(x["matches"]), int(x["fragments"]))
            k += 1
    return out


def write_fastani_file(path: Path, query: Path, reference: Path, result: Optional[ComparisonResult]) -> Path:
    """One result file as fastANI writes it (`<query>_vs_<ref>.fastani`, fastani.py:215): a single tab-separated line
    query, reference, ANI in PERCENT, matches, fragments — or an empty file when there is no result."""
    path = Path(path)
    with open(path, "w") as fh:
        if result is not None:
            fh.write(f"{query}\t{reference}\t{100.0 * result.ani:.4f}\t{result.matches}\t{result.fragments}\n")
    return path


def parse_fastani_file(filename: Path) -> ComparisonResult:
    """fastani.py:231-270: the first line of a result file -> ComparisonResult(column 0, column 1, 0.01 * column 2, int, int) (the
    reference passes the columns through in file order: query file first); an empty file raises PyaniFastANIException."""
    with open(filename, "r") as fh:
        line = fh.readline().strip().split()
    if not line:
        raise PyaniFastANIException(f"Input file {filename} is empty")
    return ComparisonResult(line[0], line[1], 0.01 * float(line[2]), int(line[3]), int(line[4]))


def comparison_row(result: Optional[ComparisonResult], query: Path, reference: Path, fragLen: int, query_length: int) -> dict:
    """The Comparison row the reference's driver makes of one result (subcmd_fastani.py:437-474): an empty result file becomes
    (query, ref, 0, 0, 0); aln_length = matches * fragLen, sim_errs = (fragments - matches) * fragLen, cov_query = matches * fragLen /
    query length, identity = the ANI fraction, cov_subject None."""
    if result is None:
        result = ComparisonResult(query, reference, 0, 0, 0)
    q, r, ani, matches, num_frags = result
    return {"query": q, "subject": r, "aln_length": int(matches * fragLen), "sim_errs": int(int(num_frags) * fragLen - matches * fragLen),
            "identity": float(ani), "cov_query": float(matches) * fragLen / query_length, "cov_subject": None, "program": "fastANI",
            "fragsize": fragLen, "maxmatch": False}


def result_matrices(labels: Sequence[str], results: Dict[Tuple[str, str], Optional[ComparisonResult]]):
    """Query-by-reference matrices of a sketch run as plain dict-of-dicts (rows = query): `identity` (the ANI estimate, NaN where
    there is no result), `matches`, `fragments`, `coverage` = matches / fragments — own columns, never merged with ANIm's."""
    nan = float("nan")
    ident = {a: {b: nan for b in labels} for a in labels}
    matches = {a: {b: 0 for b in labels} for a in labels}
    frags = {a: {b: 0 for b in labels} for a in labels}
    cov = {a: {b: nan for b in labels} for a in labels}
    for (q, r), x in results.items():
        if x is None:
            continue
        ident[q][r] = x.ani
        matches[q][r], frags[q][r] = x.matches, x.fragments
        cov[q][r] = x.matches / x.fragments if x.fragments else nan
    return {"identity": ident, "matches": matches, "fragments": frags, "coverage": cov}
