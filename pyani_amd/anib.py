"""ANIb pieces of the hot path — what pyani itself computes around BLAST (pyani/anib.py).

    fragment_lengths / fragment_records   the 1020-nt fragmenting rule          (anib.py:164-203, FRAGSIZE pyani_config.py:95)
    parse_blast_tab(filename)             (aln_length, sim_errors, mean pident)  (anib.py:569-667, mode "ANIb"), reduced on the GPU
    process_blast_results(...)            identity / coverage / lengths / errors / hadamard matrices, [q, s] cells only
                                          (process_blast, anib.py:496-565)

The fragment-vs-genome SEARCH (BLAST+ `blastn -task blastn`, external, absent from the reference tree) is not
implemented yet: fragment mode of the GPU aligner is the next §8 row.  Until then this module reduces existing
`.blast_tab` files exactly as pyani does.
"""
import gzip
from pathlib import Path
from typing import Dict, Iterable, List, Tuple

import numpy as np
import pandas as pd

from .engine import Engine, default_engine

FRAGSIZE = 1020  # pyani_config.FRAGSIZE


def fragment_lengths(record_lengths: Iterable[int], fragsize: int = FRAGSIZE) -> Dict[str, int]:
    """Fragment ids and lengths for one FASTA file: every record is cut at 0, fragsize, 2*fragsize, ...; the last piece
    may be shorter; ids frag00001... run across all records of the file (anib.py:190-200)."""
    out, count = {}, 0
    for n in record_lengths:
        idx = 0
        while idx < n:
            count += 1
            out["frag%05d" % count] = min(fragsize, n - idx)
            idx += fragsize
    return out


def fragment_records(records: Iterable[Tuple[str, str]], fragsize: int = FRAGSIZE) -> List[Tuple[str, str]]:
    """(title, sequence) records -> [(frag id, sequence piece)] with the same rule."""
    out, count = [], 0
    for _, seq in records:
        for idx in range(0, len(seq), fragsize):
            count += 1
            out.append(("frag%05d" % count, seq[idx: idx + fragsize]))
    return out


def read_blast_tab(path):
    """15-column BLAST+ table -> (n_frags, rows) for Engine.anib_reduce; fragment ordinals follow sorted id order."""
    opener = gzip.open if str(path).endswith(".gz") else open
    raw = []
    with opener(path, "rt") as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) >= 15:
                raw.append((f[0], int(f[2]), int(f[3]), int(f[14]), int(f[6]), float(f[4])))
    ids = {q: k for k, q in enumerate(sorted({r[0] for r in raw}))}
    return len(ids), [(ids[r[0]],) + r[1:] for r in raw]


def parse_blast_tab(filename, engine: Engine = None) -> Tuple[int, int, float]:
    """Return (alignment length, similarity errors, mean_pid) of a BLAST+ .blast_tab file (anib.py:569-667)."""
    eng = engine or default_engine()
    aln, err, pid = eng.anib_reduce([read_blast_tab(filename)])
    return int(aln[0]), int(err[0]), float(pid[0])


def process_blast_results(pair_results: Dict[Tuple[str, str], Tuple[int, int, float]], org_lengths: Dict[str, int]
                          ) -> Dict[str, pd.DataFrame]:
    """process_blast's matrices (anib.py:536-564): only [query, subject] cells are written; identity = 0.01 * mean pident,
    coverage = aln_length / len[query]; hadamard = identity * coverage."""
    labels = list(org_lengths)
    n = len(labels)
    lengths = pd.DataFrame(np.full((n, n), np.nan), index=labels, columns=labels)
    errors = pd.DataFrame(np.zeros((n, n)), index=labels, columns=labels)
    ident = pd.DataFrame(np.ones((n, n)), index=labels, columns=labels)
    cov = pd.DataFrame(np.ones((n, n)), index=labels, columns=labels)
    for g in labels:
        lengths.loc[g, g] = org_lengths[g]
    for (q, s), (aln, err, pid) in pair_results.items():
        lengths.loc[q, s] = aln
        errors.loc[q, s] = err
        ident.loc[q, s] = 0.01 * pid
        cov.loc[q, s] = float(aln) / org_lengths[q]
    return {"alignment_lengths": lengths, "similarity_errors": errors, "percentage_identity": ident,
            "alignment_coverage": cov, "hadamard": ident * cov}
