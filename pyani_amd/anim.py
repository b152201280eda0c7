"""ANIm on the MI355X — in-process replacement for pyani's nucmer/delta-filter job path (pyani/anim.py).

The reference builds shell commands (`construct_nucmer_cmdline`, anim.py:240-289), runs them as OS processes
(`run_multiprocessing.run_dependency_graph`), then parses each `.filter` file (`parse_delta`, anim.py:292-411) and
assembles matrices (`process_deltadir`, anim.py:415-497; v0.3: `update_comparison_matrices`, pyani_orm.py:618-666).
Here the per-pair 4-tuples come straight from the GPU engine (`pg_anim_pairs`); the functions below keep the
reference's names, argument meaning, result containers and error behaviour for everything downstream of the jobs:

    calculate_anim_pairs(infiles)         all N(N-1) ordered comparisons -> {(qstem, sstem): (ref_aln, qry_aln, id, errs)}
    parse_delta(filename)                 same 4-tuple from an existing MUMmer .delta/.filter file (GPU reduction;
                                          pyani's --recovery path)
    process_deltadir(delta_dir, lengths)  the reference's directory walk over `*/*.filter` files, all files reduced in one GPU call
    process_deltadir-equivalent           `assemble_legacy_results(pair_results, org_lengths)` -> ANIResults
    update_comparison_matrices-equivalent `assemble_run_matrices(pair_results, lengths)` -> 5 DataFrames (v0.3 semantics)

The alignment search emulates MUMmer 3.23 (`nucmer --mum`, `delta-filter -1`), which is NOT part of the reference
tree: it is calibrated against the MUMmer output files the reference's tests hold (all 17 fixture pairs with
FASTA inputs are reproduced bit for bit; beyond the fixtures parity is unpinned).  `program`/`version` strings for DB rows
must therefore differ from "nucmer" (SURVEY.md §5): use PROGRAM / VERSION below.
"""
import gzip
import logging
import sys
from pathlib import Path
from typing import Dict, Iterable, List, Tuple

import numpy as np
import pandas as pd

from . import __version__, _lib
from .engine import Engine, default_engine

PROGRAM = "pyani_amd-anim"
VERSION = f"{__version__} (gfx950; emulates nucmer 3.1 --mum + delta-filter -1)"


class PyaniANImException(Exception):
    """ANIm-specific exception (mirrors pyani.anim.PyaniANImException)."""


class ANIResults:
    """Container mirroring pyani.pyani_tools.ANIResults (pyani_tools.py:85-196) for the ANIm mode."""

    def __init__(self, labels: List[str], mode: str = "ANIm"):
        self.alignment_lengths = pd.DataFrame(index=labels, columns=labels, dtype=float)
        self.similarity_errors = pd.DataFrame(index=labels, columns=labels, dtype=float).fillna(0)
        self.percentage_identity = pd.DataFrame(index=labels, columns=labels, dtype=float).fillna(1.0)
        self.alignment_coverage = pd.DataFrame(index=labels, columns=labels, dtype=float).fillna(1.0)
        self.zero_error = False
        self.mode = mode

    def add_tot_length(self, qname, sname, qlen, slen=None, sym=True):
        self.alignment_lengths.loc[qname, sname] = qlen
        if sym and slen:
            self.alignment_lengths.loc[sname, qname] = slen

    def add_sim_errors(self, qname, sname, value, sym=True):
        self.similarity_errors.loc[qname, sname] = value
        if sym:
            self.similarity_errors.loc[sname, qname] = value

    def add_pid(self, qname, sname, value, sym=True):
        self.percentage_identity.loc[qname, sname] = value

    def add_coverage(self, qname, sname, qcover, scover=None):
        self.alignment_coverage.loc[qname, sname] = qcover
        if scover:
            self.alignment_coverage.loc[sname, qname] = scover

    @property
    def hadamard(self) -> pd.DataFrame:
        return self.alignment_coverage * self.percentage_identity

    @property
    def data(self):
        stems = ("ANIm_alignment_lengths", "ANIm_percentage_identity", "ANIm_alignment_coverage",
                 "ANIm_similarity_errors", "ANIm_hadamard")
        return list(zip((self.alignment_lengths, self.percentage_identity, self.alignment_coverage,
                         self.similarity_errors, self.hadamard), stems))


def _tuple(rec) -> Tuple[int, int, float, int]:
    if rec["status"] == _lib.PG_ANIM_NO_ALIGNMENT:
        raise ZeroDivisionError("division by zero")            # what parse_delta raises on an empty .filter (anim.py:396)
    if rec["status"] != 0:
        raise PyaniANImException(f"GPU ANIm comparison failed with status {int(rec['status'])}")
    return int(rec["ref_aln_len"]), int(rec["qry_aln_len"]), float(rec["identity"]), int(rec["sim_errors"])


def calculate_anim_pairs(infiles: Iterable, engine: Engine = None, nofilter: bool = False, skip_zero: bool = False,
                         maxmatch: bool = False
                         ) -> Tuple[Dict[Tuple[str, str], Tuple[int, int, float, int]], Dict[str, int]]:
    """All ordered comparisons between the FASTA files (what generate_nucmer_jobs + run_dependency_graph + parse_delta
    produce, anim.py:155-235).  Key (q, s): q is nucmer's reference / pyani's query genome.  Returns (results,
    genome lengths keyed by stem)."""
    eng = engine or default_engine()
    files = sorted(Path(f) for f in infiles)
    scratch_store = eng.genome_count() == 0
    ids, lengths = {}, {}
    for f, (gid, total, _) in zip(files, eng.add_fasta_batch(files)):
        ids[f.stem], lengths[f.stem] = gid, total
    stems = [f.stem for f in files]
    pairs = [(a, b) for a in stems for b in stems if a != b]
    recs = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs], filter_1to1=not nofilter, maxmatch=maxmatch)
    if scratch_store:
        eng.clear_genomes()
    out = {}
    for (a, b), rec in zip(pairs, recs):
        try:
            out[(a, b)] = _tuple(rec)
        except ZeroDivisionError:
            if not skip_zero:
                raise
    return out, lengths


def read_delta(path):
    """MUMmer .delta/.filter -> list of (rseq, qseq, rs, re, qs, qe, errors) with per-file sequence ordinals."""
    recs, rid, qid = [], {}, {}
    cur = None
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as fh:
        for line in fh:
            f = line.split()
            if not f or f[0] == "NUCMER":
                continue
            if f[0].startswith(">"):
                cur = (rid.setdefault(f[0][1:], len(rid)), qid.setdefault(f[1], len(qid)))
            elif len(f) == 7 and cur is not None:
                recs.append((cur[0], cur[1], int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4])))
    return recs


def fasta_records(path) -> List[Tuple[str, int]]:
    """(record id, length) of every record of a FASTA file, in file order (ids = first whitespace token, as nucmer)."""
    out = []
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as fh:
        for line in fh:
            if line.startswith(">"):
                out.append([line[1:].split()[0] if line[1:].split() else "", 0])
            elif out:
                out[-1][1] += len(line.strip().replace(" ", ""))
    return [(i, n) for i, n in out]


def write_delta(path, ref_fasta, qry_fasta, alignments, filtered: bool = True) -> int:
    """Write one pair's alignment records (Engine.anim_pair_alignments) as a MUMmer .filter (filtered=True: only the
    records delta-filter -1 keeps) or .delta file: the file nucmer / delta_filter_wrapper leave in
    <outdir>/nucmer_output/<stem1>/<stem1>_vs_<stem2>.{filter,delta} (anim.py:271-288) and pyani's --recovery mode
    reads back with parse_delta.  Header lines follow pyani/nucmer.py:292-351.  The indel offset lists of a real
    .delta are NOT written (no traceback in the engine; parse_delta ignores them): every alignment is its 7-field
    header followed by the terminating 0, so the file is exact for pyani and for coordinate-level tools, but tools that
    replay the alignment (show-aligns, dnadiff's SNP calls) cannot use it.  Returns the number of records written."""
    rrec, qrec = fasta_records(ref_fasta), fasta_records(qry_fasta)
    blocks = {}
    for a in alignments:
        if filtered and int(a["kept"]) != 3:
            continue
        blocks.setdefault((int(a["ref_rec"]), int(a["qry_rec"])), []).append(a)
    n = 0
    with open(path, "w") as fh:
        fh.write(f"{Path(ref_fasta).resolve()} {Path(qry_fasta).resolve()}\nNUCMER\n")
        for (r, q) in sorted(blocks):
            fh.write(f">{rrec[r][0]} {qrec[q][0]} {rrec[r][1]} {qrec[q][1]}\n")
            for a in sorted(blocks[(r, q)], key=lambda x: (int(x["rs"]), int(x["qs"]))):
                e = int(a["errors"])
                fh.write(f"{int(a['rs'])} {int(a['re'])} {int(a['qs'])} {int(a['qe'])} {e} {e} 0\n0\n")
                n += 1
    return n


def parse_delta(filename, engine: Engine = None) -> Tuple[int, int, float, int]:
    """(reference alignment length, query alignment length, average identity, similarity errors) of a MUMmer
    .delta/.filter file — pyani.anim.parse_delta (anim.py:292-411), reduced on the GPU."""
    eng = engine or default_engine()
    return _tuple(eng.anim_reduce([read_delta(filename)], apply_filter=False)[0])


def process_deltadir(delta_dir, org_lengths: Dict[str, int], logger=None, engine: Engine = None) -> ANIResults:
    """pyani.anim.process_deltadir (anim.py:415-497): ANIResults from the `<delta_dir>/*/*.filter` files of an earlier
    (MUMmer or write_delta) run — same file order, same skipping of files whose organisms are not in `org_lengths`, same
    overwrite order of the mirrored cells, PyaniANImException when the directory holds no .filter file, ZeroDivisionError
    from an empty one.  The files are parsed on the host and reduced in ONE GPU call instead of one parse_delta each."""
    logger = logger or logging.getLogger(__name__)
    delta_dir = Path(delta_dir)
    deltafiles = sorted(delta_dir.glob("*/*.filter"))
    logger.info("%s has %d files to load", delta_dir, len(deltafiles))
    if not deltafiles:
        logger.error("%s empty? No filter files found", delta_dir)
        raise PyaniANImException(f"{delta_dir} contains no filter files.")
    todo = []
    for deltafile in deltafiles:
        qname, sname = deltafile.stem.split("_vs_")
        if qname not in org_lengths:
            logger.warning("Query name %s not in input sequence list, skipping %s", qname, deltafile)
            continue
        if sname not in org_lengths:
            logger.warning("Subject name %s not in input sequence list, skipping %s", sname, deltafile)
            continue
        todo.append((qname, sname, deltafile))
    results = ANIResults(list(org_lengths.keys()), "ANIm")
    for org, length in org_lengths.items():
        results.alignment_lengths.loc[org, org] = length
    if not todo:
        return results
    eng = engine or default_engine()
    recs = eng.anim_reduce([read_delta(f) for _, _, f in todo], apply_filter=False)
    for (qname, sname, deltafile), rec in zip(todo, recs):
        query_tot_length, subject_tot_length, weighted_identity, tot_sim_error = _tuple(rec)
        if subject_tot_length == 0:
            logger.warning("Total alignment length reported in %s is zero!", deltafile)
            sys.exit("Zero length alignment!")
        results.add_tot_length(qname, sname, query_tot_length, subject_tot_length)
        results.add_sim_errors(qname, sname, tot_sim_error)
        results.add_pid(qname, sname, weighted_identity)
        results.add_coverage(qname, sname, float(query_tot_length) / org_lengths[qname],
                             float(subject_tot_length) / org_lengths[sname])
    return results


def assemble_legacy_results(pair_results: Dict[Tuple[str, str], Tuple[int, int, float, int]], org_lengths: Dict[str, int]
                            ) -> ANIResults:
    """process_deltadir's assembly (anim.py:438-497): files visited in sorted `<q>/<q>_vs_<s>.filter` order, mirrored
    cells overwritten by later files, identity per direction, diagonal of the length matrix = genome length."""
    results = ANIResults(list(org_lengths.keys()), "ANIm")
    for org, length in org_lengths.items():
        results.alignment_lengths.loc[org, org] = length
    for (q, s) in sorted(pair_results, key=lambda k: f"{k[0]}/{k[0]}_vs_{k[1]}.filter"):
        if q not in org_lengths or s not in org_lengths:
            continue
        qtot, stot, ident, err = pair_results[(q, s)]
        results.add_tot_length(q, s, qtot, stot)
        results.add_sim_errors(q, s, err)
        results.add_pid(q, s, ident)
        results.add_coverage(q, s, float(qtot) / org_lengths[q], float(stot) / org_lengths[s])
    return results


def assemble_run_matrices(pair_results: Dict[Tuple[str, str], Tuple[int, int, float, int]], lengths: Dict[str, int]
                          ) -> Dict[str, pd.DataFrame]:
    """v0.3 semantics of update_comparison_matrices (pyani_orm.py:618-666): [q, s] cells only, diagonals 1 / 1 /
    length / 0 / 1, hadamard = identity * cov_query.  Vectorised: the reference fills the five
    matrices with one pandas scalar write per cell, which at N = 1000 takes longer than the GPU needs for the pairs."""
    labels = sorted(lengths)
    n = len(labels)
    idx = {g: k for k, g in enumerate(labels)}
    length = np.array([lengths[g] for g in labels], dtype=np.float64)
    m = len(pair_results)
    qi = np.fromiter((idx[q] for q, _ in pair_results), dtype=np.int64, count=m)
    si = np.fromiter((idx[s] for _, s in pair_results), dtype=np.int64, count=m)
    vals = np.array(list(pair_results.values()), dtype=np.float64).reshape(m, 4)   # (qaln, saln, identity, errors); ints < 2^53
    ident, cov, had = np.eye(n), np.eye(n), np.eye(n)
    aln, sim = np.zeros((n, n)), np.zeros((n, n))
    np.fill_diagonal(aln, length)
    cq = vals[:, 0] / length[qi]
    ident[qi, si] = vals[:, 2]
    cov[qi, si] = cq
    aln[qi, si] = vals[:, 0]
    sim[qi, si] = vals[:, 3]
    had[qi, si] = vals[:, 2] * cq

    def frame(a):
        return pd.DataFrame(a, index=labels, columns=labels)

    return {"identity": frame(ident), "coverage": frame(cov), "aln_lengths": frame(aln), "sim_errors": frame(sim),
            "hadamard": frame(had)}
