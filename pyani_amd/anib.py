"""ANIb pieces of the hot path — what pyani itself computes around BLAST (pyani/anib.py).

    fragment_lengths / fragment_records   the 1020-nt fragmenting rule          (anib.py:164-203, FRAGSIZE pyani_config.py:95)
    fragment_fasta_files / get_fraglength_dict / get_fragment_lengths   the same rule at file level, names and returns as pyani
    parse_blast_tab(filename)             (aln_length, sim_errors, mean pident)  (anib.py:569-667, mode "ANIb"), reduced on the GPU
    process_blast_results(...)            identity / coverage / lengths / errors / hadamard matrices, [q, s] cells only
                                          (process_blast, anib.py:496-565)

    calculate_anib_pairs(infiles)         all N(N-1) ordered comparisons on the GPU: fragment search + reduction
                                          (what generate_blastn_commands + the BLAST jobs + parse_blast_tab produce, anib.py:383-667)
    anib_pair_table / write_blast_tab     one pair's table in BLAST+'s 15-column layout (anib.py:465-471), which pyani's own
                                          parse_blast_tab reads back

The fragment-vs-genome SEARCH replaces BLAST+ (`blastn -task blastn`, third-party, absent from the reference tree) with the
engine's fragment mode (pg_anib_pairs: seeds from the LDS-table seeding, X-drop gapped extension with blastn's scores);
how closely it follows BLAST+'s own tables on the reference's fixtures is stated in DESIGN.md.
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


def _read_fasta(path) -> List[Tuple[str, str]]:
    """[(header line without '>', sequence)] of a (possibly gzipped) FASTA file; blank lines ignored."""
    opener = gzip.open if str(path).endswith(".gz") else open
    out, title, parts = [], None, []
    with opener(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if title is not None:
                    out.append((title, "".join(parts)))
                title, parts = line[1:], []
            elif title is not None and line:
                parts.append(line.strip())
    if title is not None:
        out.append((title, "".join(parts)))
    return out


def fragment_fasta_files(infiles: List[Path], outdirname: Path, fragsize: int = FRAGSIZE) -> Tuple[List[Path], Dict]:
    """pyani.anib.fragment_fasta_files (anib.py:164-203): every sequence of every file cut into consecutive pieces of
    `fragsize` (the last one may be shorter), written to `<outdirname>/<stem>-fragments<suffix>` with ids fragNNNNN
    running across the records of a file and the original header kept as description (Biopython's FASTA writer:
    `>fragNNNNN <original header>`, 60 columns).  Returns (filenames, {stem: {frag id: length}})."""
    outdirname = Path(outdirname)
    outfnames = []
    for fname in (Path(f) for f in infiles):
        outfname = outdirname / f"{fname.stem}-fragments{fname.suffix}"
        count = 0
        with open(outfname, "w") as fh:
            for title, seq in _read_fasta(fname):
                for idx in range(0, len(seq), fragsize):
                    count += 1
                    piece = seq[idx: idx + fragsize]
                    fh.write(f">frag{count:05d} {title}\n")
                    for col in range(0, len(piece), 60):
                        fh.write(piece[col: col + 60] + "\n")
        outfnames.append(outfname)
    return outfnames, get_fraglength_dict(outfnames)


def get_fraglength_dict(fastafiles: List[Path]) -> Dict[str, Dict[str, int]]:
    """pyani.anib.get_fraglength_dict (anib.py:207-221): fragment lengths per file, keyed by the name before '-fragments'."""
    return {Path(f).stem.split("-fragments")[0]: get_fragment_lengths(f) for f in fastafiles}


def get_fragment_lengths(fastafile: Path) -> Dict[str, int]:
    """pyani.anib.get_fragment_lengths (anib.py:224-238): sequence lengths keyed by sequence id (first word of the header);
    ambiguity symbols are not discounted."""
    return {(title.split(None, 1)[0] if title.split() else ""): len(seq) for title, seq in _read_fasta(fastafile)}


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


def calculate_anib_pairs(infiles: Iterable, engine: Engine = None, fragsize: int = FRAGSIZE
                         ) -> Tuple[Dict[Tuple[str, str], Tuple[int, int, float]], Dict[str, int]]:
    """All ordered comparisons between the FASTA files: {(query stem, subject stem): (aln_length, sim_errors, mean pident)}
    — parse_blast_tab's tuple for `<query>_vs_<subject>.blast_tab` — and the genome lengths keyed by stem."""
    eng = engine or default_engine()
    files = sorted(Path(f) for f in infiles)
    stems = [f.stem for f in files]
    if len(set(stems)) != len(stems):
        raise ValueError("two input files share a stem (pyani keys every result by Path.stem)")
    scratch_store = eng.genome_count() == 0
    ids, lengths = {}, {}
    try:
        for f, (gid, total, _) in zip(files, eng.add_fasta_batch(files)):
            ids[f.stem], lengths[f.stem] = gid, total
        pairs = [(a, b) for a in stems for b in stems if a != b]
        recs = eng.anib_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs], fragsize)
    finally:
        if scratch_store:
            eng.clear_genomes()
    out = {}
    for (a, b), r in zip(pairs, recs):
        if int(r["status"]) != 0:
            raise RuntimeError(f"GPU ANIb comparison {a} vs {b} failed with status {int(r['status'])}")
        out[(a, b)] = (int(r["aln_length"]), int(r["sim_errors"]), float(r["pid"]))
    return out, lengths


def write_blast_tab(path, rows, subject_ids: List[str], subject_lengths: List[int]) -> int:
    """Engine.anib_pair_rows -> a `.blast_tab` file in the column layout pyani asks BLAST+ for (anib.py:465-471):
    qseqid sseqid length mismatch pident nident qlen slen qstart qend sstart send positive ppos gaps.
    subject_ids / subject_lengths: id and length of every record of the subject FASTA.  Returns the number of rows."""
    with open(path, "w") as fh:
        for r in rows:
            pid = 100.0 * int(r["nident"]) / int(r["length"])
            rec = int(r["srec"])
            fh.write("\t".join(str(x) for x in (
                "frag%05d" % (int(r["frag"]) + 1), subject_ids[rec], int(r["length"]), int(r["mismatch"]), "%.3f" % pid, int(r["nident"]),
                int(r["qlen"]), subject_lengths[rec], int(r["qstart"]), int(r["qend"]), int(r["sstart"]), int(r["send"]),
                int(r["nident"]), "%.2f" % pid, int(r["gaps"]))) + "\n")
    return len(rows)


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
