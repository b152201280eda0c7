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
tree: it is pinned twice — against the MUMmer output files the reference's tests hold (all 43 fixture runs with FASTA inputs,
25 192 records, bit for bit) and, beyond the fixtures, against an independent restatement of MUMmer 3.23's own algorithms
(the benchmark-scale golden records under tests/golden/, filter ON and OFF).  `program`/`version` strings for DB rows
must therefore differ from "nucmer" (SURVEY.md §5): use PROGRAM / VERSION below.
"""
import gzip
import logging
import sys
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import pandas as pd

from . import __version__, _lib
from .engine import Engine, default_engine

PROGRAM = "pyani_amd-anim"
VERSION = f"{__version__} (gfx950; emulates nucmer 3.1 --mum + delta-filter -1)"


class PyaniANImException(Exception):
    """ANIm-specific exception (mirrors pyani.anim.PyaniANImException)."""


class ANIResults:
    """The five result matrices of a run, under the attribute names pyani's callers read (pyani_tools.py:85-196).  Filled
    whole by `assemble_legacy_results` (vectorised, last writer wins as in the reference's per-cell loop); there are no
    per-cell setters."""

    def __init__(self, labels: List[str], mode: str = "ANIm"):
        self.alignment_lengths = pd.DataFrame(index=labels, columns=labels, dtype=float)
        self.similarity_errors = pd.DataFrame(index=labels, columns=labels, dtype=float).fillna(0)
        self.percentage_identity = pd.DataFrame(index=labels, columns=labels, dtype=float).fillna(1.0)
        self.alignment_coverage = pd.DataFrame(index=labels, columns=labels, dtype=float).fillna(1.0)
        self.zero_error = False
        self.mode = mode

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
                         maxmatch: bool = False, devices=None, workers=None
                         ) -> Tuple[Dict[Tuple[str, str], Tuple[int, int, float, int]], Dict[str, int]]:
    """All ordered comparisons between the FASTA files (what generate_nucmer_jobs + run_dependency_graph + parse_delta
    produce, anim.py:155-235).  Key (q, s): q is nucmer's reference / pyani's query genome.  Returns (results,
    genome lengths keyed by stem).  devices / workers: several GPUs of this node through a work queue (pyani_amd/multi.py)."""
    if engine is None and (devices is not None or workers):
        from . import multi
        eng = multi.engine_for(devices, workers)
        try:
            return calculate_anim_pairs(infiles, eng, nofilter, skip_zero, maxmatch)
        finally:
            if isinstance(eng, multi.MultiEngine):
                eng.close()
    eng = engine or default_engine()
    files = sorted(Path(f) for f in infiles)
    stems = [f.stem for f in files]
    if len(set(stems)) != len(stems):
        raise ValueError("two input files share a stem (pyani keys every result by Path.stem): "
                         + ", ".join(sorted({s for s in stems if stems.count(s) > 1})))
    scratch_store = eng.genome_count() == 0
    ids, lengths = {}, {}
    try:
        for f, (gid, total, _) in zip(files, eng.add_fasta_batch(files)):
            ids[f.stem], lengths[f.stem] = gid, total
        pairs = [(a, b) for a in stems for b in stems if a != b]
        recs = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs], filter_1to1=not nofilter, maxmatch=maxmatch)
    finally:
        if scratch_store:       # also on errors: a failed call must not leave its genomes in the shared engine
            eng.clear_genomes()
    out = {}
    for (a, b), rec in zip(pairs, recs):
        try:
            out[(a, b)] = _tuple(rec)
        except ZeroDivisionError:
            if not skip_zero:
                raise
    return out, lengths


def read_delta(path, with_indels: bool = False):
    """MUMmer .delta/.filter -> list of (rseq, qseq, rs, re, qs, qe, errors) with per-file sequence ordinals; with_indels=True:
    (records, indel lists) — record k's signed indel offsets without the terminating 0 (pyani_amd.nucmer.DeltaData is the object
    model of the same files, pyani/nucmer.py:47-351)."""
    if with_indels:
        from .nucmer import DeltaData
        recs, lists, rid, qid = [], [], {}, {}
        for comp in DeltaData.from_file(path).comparisons:
            cur = (rid.setdefault(comp.header.reference, len(rid)), qid.setdefault(comp.header.query, len(qid)))
            for a in comp.alignments:
                recs.append((cur[0], cur[1], a.refstart, a.refend, a.querystart, a.queryend, a.errs))
                lists.append(a.indel_offsets)
        return recs, lists
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


def write_delta(path, ref_fasta, qry_fasta, alignments, filtered: bool = True, indels=None) -> int:
    """Write one pair's alignment records (Engine.anim_alignments_batch / anim_pair_alignments) as a MUMmer .filter
    (filtered=True: only the records delta-filter -1 keeps) or .delta file: the file nucmer / delta_filter_wrapper leave in
    <outdir>/nucmer_output/<stem1>/<stem1>_vs_<stem2>.{filter,delta} (anim.py:271-288) and pyani's --recovery mode
    reads back with parse_delta.  Header lines follow pyani/nucmer.py:292-351.
    indels: one sequence of indel offsets per record (anim_alignments_batch(with_indels=True)) — then every alignment header is
    followed by its offsets and the terminating 0, as in MUMmer's own file, and the records keep the order they came in
    (MUMmer's output order inside each sequence pair).  Without them every alignment is its header followed by the 0 alone (exact
    for pyani, whose parse_delta ignores the lists; tools that replay alignments need them) and records are sorted by start.
    Returns the number of records written."""
    rrec, qrec = fasta_records(ref_fasta), fasta_records(qry_fasta)
    blocks = {}
    for k, a in enumerate(alignments):
        if filtered and int(a["kept"]) != 3:
            continue
        blocks.setdefault((int(a["ref_rec"]), int(a["qry_rec"])), []).append((a, None if indels is None else indels[k]))
    n = 0
    with open(path, "w") as fh:
        fh.write(f"{Path(ref_fasta).resolve()} {Path(qry_fasta).resolve()}\nNUCMER\n")
        for (r, q) in sorted(blocks):
            fh.write(f">{rrec[r][0]} {qrec[q][0]} {rrec[r][1]} {qrec[q][1]}\n")
            recs = blocks[(r, q)] if indels is not None else sorted(blocks[(r, q)], key=lambda x: (int(x[0]["rs"]), int(x[0]["qs"])))
            for a, ind in recs:
                e = int(a["errors"])
                fh.write(f"{int(a['rs'])} {int(a['re'])} {int(a['qs'])} {int(a['qe'])} {e} {e} 0\n")
                if ind is not None and len(ind):
                    fh.write("\n".join(str(int(v)) for v in ind) + "\n")
                fh.write("0\n")
                n += 1
    return n


def parse_delta(filename, engine: Engine = None) -> Tuple[int, int, float, int]:
    """(reference alignment length, query alignment length, average identity, similarity errors) of a MUMmer
    .delta/.filter file — pyani.anim.parse_delta (anim.py:292-411), reduced on the GPU."""
    eng = engine or default_engine()
    return _tuple(eng.anim_reduce([read_delta(filename)], apply_filter=False)[0])


def process_deltadir(delta_dir, org_lengths: Dict[str, int], logger=None, engine: Engine = None) -> ANIResults:
    """pyani.anim.process_deltadir (anim.py:415-497) for the `<delta_dir>/*/*.filter` files of an earlier (MUMmer or
    write_delta) run: glob -> read_delta -> ONE batched GPU reduction -> assemble_legacy_results.  Behaviour kept: files
    in sorted path order (it decides which file's value a mirrored cell ends up with), files naming an organism that is
    not in `org_lengths` are skipped with a warning, PyaniANImException when there is no .filter file at all,
    ZeroDivisionError from a file without alignments, exit on a zero subject length."""
    log = logger or logging.getLogger(__name__)
    files = sorted(Path(delta_dir).glob("*/*.filter"))
    log.info("%s: %d .filter files", delta_dir, len(files))
    if not files:
        raise PyaniANImException(f"{delta_dir} contains no filter files.")
    keep = []
    for f in files:
        q, s = f.stem.split("_vs_")
        missing = [name for name in (q, s) if name not in org_lengths]
        if missing:
            log.warning("%s: %s not among the input sequences, file skipped", f, " / ".join(missing))
        else:
            keep.append((q, s, f))
    pair_results = {}
    if keep:
        eng = engine or default_engine()
        for (q, s, f), rec in zip(keep, eng.anim_reduce([read_delta(f) for _, _, f in keep], apply_filter=False)):
            pair_results[(q, s)] = _tuple(rec)
            if pair_results[(q, s)][1] == 0:
                log.warning("%s reports a total alignment length of zero", f)
                sys.exit("Zero length alignment!")
    return assemble_legacy_results(pair_results, org_lengths, order=[(q, s) for q, s, _ in keep])


def _last_writer_wins(out: np.ndarray, cells: np.ndarray, ranks: np.ndarray, values: np.ndarray) -> None:
    """out.flat[cell] = the value of the write with the highest rank among those addressed to that cell."""
    if len(cells) == 0:
        return
    by = np.lexsort((ranks, cells))
    c = cells[by]
    last = np.r_[c[1:] != c[:-1], True]
    out.flat[c[last]] = values[by][last]


def assemble_legacy_results(pair_results: Dict[Tuple[str, str], Tuple[int, int, float, int]], org_lengths: Dict[str, int],
                            order: Optional[List[Tuple[str, str]]] = None) -> ANIResults:
    """The matrices process_deltadir builds (anim.py:438-497, through ANIResults.add_*, pyani_tools.py:108-167), from the
    per-pair tuples.  Every file (q, s) writes its own cell [q, s] of all four matrices AND the mirrored cell [s, q] of
    the length, error and coverage matrices (length / coverage only when non-zero), so a cell ends up with whatever the
    LAST file that touched it wrote: `order` is that file order (default: sorted `<q>/<q>_vs_<s>.filter` paths, which
    compare component by component).  Identity is written per direction only; the length diagonal is the genome length.
    Vectorised: each matrix is resolved with one last-writer-wins scatter instead of 7 pandas scalar writes per file."""
    labels = list(org_lengths)
    n = len(labels)
    pos = {g: k for k, g in enumerate(labels)}
    if order is None:
        order = sorted(pair_results, key=lambda k: (k[0], f"{k[0]}_vs_{k[1]}.filter"))
    order = [k for k in order if k[0] in pos and k[1] in pos]
    m = len(order)
    qi = np.fromiter((pos[q] for q, _ in order), dtype=np.int64, count=m)
    si = np.fromiter((pos[s] for _, s in order), dtype=np.int64, count=m)
    vals = np.array([pair_results[k] for k in order], dtype=np.float64).reshape(m, 4)
    rank = np.arange(m, dtype=np.int64)
    length = np.array([org_lengths[g] for g in labels], dtype=np.float64)
    own, mirror = qi * n + si, si * n + qi
    aln = np.full((n, n), np.nan)
    np.fill_diagonal(aln, length)
    err, pid, cov = np.zeros((n, n)), np.ones((n, n)), np.ones((n, n))
    qcov, scov = vals[:, 0] / length[qi] if m else vals[:, 0], vals[:, 1] / length[si] if m else vals[:, 1]
    has_s, has_scov = vals[:, 1] != 0, scov != 0          # add_tot_length / add_coverage mirror only truthy values
    _last_writer_wins(aln, np.r_[own, mirror[has_s]], np.r_[rank, rank[has_s]], np.r_[vals[:, 0], vals[has_s, 1]])
    _last_writer_wins(err, np.r_[own, mirror], np.r_[rank, rank], np.r_[vals[:, 3], vals[:, 3]])
    _last_writer_wins(pid, own, rank, vals[:, 2])
    _last_writer_wins(cov, np.r_[own, mirror[has_scov]], np.r_[rank, rank[has_scov]], np.r_[qcov, scov[has_scov]])
    results = ANIResults(labels, "ANIm")
    for name, a in (("alignment_lengths", aln), ("similarity_errors", err), ("percentage_identity", pid), ("alignment_coverage", cov)):
        setattr(results, name, pd.DataFrame(a, index=labels, columns=labels))
    return results


def assemble_run_matrices(pair_results: Dict[Tuple[str, str], Tuple[int, int, float, int]], lengths: Dict[str, int],
                          genome_ids: Optional[Dict[str, int]] = None) -> Dict[str, pd.DataFrame]:
    """update_comparison_matrices (pyani_orm.py:618-666), cell for cell: five float frames that start as NaN, diagonals
    1 / 1 / genome length / 0 / 1, then ONLY the [query, subject] cell of every comparison that exists: identity,
    cov_query = aln_length / query length, aln_length, sim_errs, identity * cov_query.  A pair that is absent (no
    alignment with skip_zero, a partial or recovered run) therefore stays NaN, as in the reference.  Index / columns:
    the sorted integer genome_ids when `genome_ids` (stem -> id) is given — what the reference stores — else the sorted
    stems.  Vectorised: the reference does one pandas scalar write per cell, ~10^7 of them at N = 1000."""
    if genome_ids is not None:
        order = sorted(lengths, key=lambda g: genome_ids[g])
        labels = [genome_ids[g] for g in order]
    else:
        order = labels = sorted(lengths)
    n = len(order)
    idx = {g: k for k, g in enumerate(order)}
    length = np.array([lengths[g] for g in order], dtype=np.float64)
    m = len(pair_results)
    qi = np.fromiter((idx[q] for q, _ in pair_results), dtype=np.int64, count=m)
    si = np.fromiter((idx[s] for _, s in pair_results), dtype=np.int64, count=m)
    vals = np.array(list(pair_results.values()), dtype=np.float64).reshape(m, 4)   # (qaln, saln, identity, errors); ints < 2^53
    ident, cov, aln, sim, had = (np.full((n, n), np.nan) for _ in range(5))
    np.fill_diagonal(ident, 1.0)
    np.fill_diagonal(cov, 1.0)
    np.fill_diagonal(aln, length)
    np.fill_diagonal(sim, 0.0)
    np.fill_diagonal(had, 1.0)
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


def run_matrices_to_json(mats: Dict[str, pd.DataFrame]) -> Dict[str, str]:
    """The five strings update_comparison_matrices stores in the Run row (pyani_orm.py:661-665): DataFrame.to_json()."""
    return {"df_identity": mats["identity"].to_json(), "df_coverage": mats["coverage"].to_json(),
            "df_alnlength": mats["aln_lengths"].to_json(), "df_simerrors": mats["sim_errors"].to_json(),
            "df_hadamard": mats["hadamard"].to_json()}


def comparison_rows(pair_results: Dict[Tuple[str, str], Tuple[int, int, float, int]], lengths: Dict[str, int],
                    genome_ids: Dict[str, int], maxmatch: bool = False, program: str = PROGRAM, version: str = VERSION
                    ) -> List[dict]:
    """One dict per comparison with the columns of the reference's Comparison table (pyani_orm.py:262-322) as
    update_comparison_results fills them for ANIm (subcmd_anim.py:434-456): aln_length = the QUERY's aligned length,
    cov_query / cov_subject = aligned length / genome length, fragsize / kmersize / minmatch NULL."""
    rows = []
    for (q, s), (qaln, saln, ident, errs) in pair_results.items():
        rows.append({"query_id": genome_ids[q], "subject_id": genome_ids[s], "aln_length": qaln, "sim_errs": errs,
                     "identity": ident, "cov_query": qaln / lengths[q], "cov_subject": saln / lengths[s], "program": program,
                     "version": version, "fragsize": None, "maxmatch": maxmatch, "kmersize": None, "minmatch": None})
    return rows
