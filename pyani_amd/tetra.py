"""TETRA on the MI355X — drop-in for the reference's ``pyani.tetra`` module API (pyani/tetra.py).

Same function names, argument meaning, return types and error behaviour as the reference:

    calculate_tetra_zscores(infilenames)  -> Dict[stem, Dict[tetranucleotide, Z]]     (pyani/tetra.py:66-74)
    calculate_tetra_zscore(filename)      -> Dict[tetranucleotide, Z]                 (pyani/tetra.py:78-139)
    tetra_clean(instr)                    -> bool                                      (pyani/tetra.py:143-153)
    calculate_correlations(tetra_z)       -> pandas.DataFrame                          (pyani/tetra.py:158-194)
    calculate_tetra(infiles)              -> pandas.DataFrame   (scripts/average_nucleotide_identity.py:582-612)

All arithmetic runs in the HIP kernels behind include/pyani_gpu.h; values are bit-identical to the reference's
(run under CPython 3.10 semantics).  One documented difference: result dicts are keyed in sorted tetranucleotide
order, the reference's in first-observation order (dict equality is unaffected).
"""
from itertools import product
from pathlib import Path
from typing import Dict, Iterable, List

import numpy as np
import pandas as pd

from . import _lib
from .engine import Engine, default_engine

TETRAMERS: List[str] = ["".join(p) for p in product("ACGT", repeat=4)]
_INDEX = {t: i for i, t in enumerate(TETRAMERS)}


def tetra_clean(instr: str) -> bool:
    """Return True if the string holds only A, C, G, T (case-sensitive, like the reference)."""
    return not (set(instr) - set("ACGT"))


def _z_to_dict(z_row: np.ndarray, present_row: np.ndarray) -> Dict[str, float]:
    return {TETRAMERS[t]: float(z_row[t]) for t in np.flatnonzero(present_row)}


def calculate_tetra_zscores(infilenames: Iterable, engine: Engine = None) -> Dict[str, Dict[str, float]]:
    """Return dictionary of TETRA Z-scores for each input file, keyed by file stem (one batched GPU pass)."""
    eng = engine or default_engine()
    files = [Path(f) for f in infilenames]
    scratch_store = eng.genome_count() == 0
    try:
        ids = [g for g, _, _ in eng.add_fasta_batch(files)]   # multithreaded read + parse + pack
        z, present, _ = eng.tetra_matrix(ids, want_corr=False)
    finally:
        if scratch_store:       # also on errors: a failed call must not leave its genomes in the shared engine
            eng.clear_genomes()
    out: Dict[str, Dict[str, float]] = {}
    for k, f in enumerate(files):
        out[f.stem] = _z_to_dict(z[k], present[k])
    return out


def calculate_tetra_zscore(filename: Path, engine: Engine = None) -> Dict[str, float]:
    """Return TETRA Z-scores for the sequence(s) in the passed FASTA file."""
    return calculate_tetra_zscores([filename], engine)[Path(filename).stem]


def calculate_correlations(tetra_z: Dict[str, Dict[str, float]], engine: Engine = None) -> pd.DataFrame:
    """Return dataframe of Pearson correlation coefficients between the organisms' Z-score vectors.

    Rows/columns are the sorted organism names, diagonal 1.0.  Raises AssertionError when two organisms have
    different tetranucleotide key sets (pyani/tetra.py:174-175) and ZeroDivisionError for empty key sets (:181).
    """
    eng = engine or default_engine()
    orgs = sorted(tetra_z.keys())
    n = len(orgs)
    z = np.zeros((n, 256), dtype=np.float64)
    present = np.zeros((n, 256), dtype=np.uint8)
    for i, org in enumerate(orgs):
        for tet, val in tetra_z[org].items():
            idx = _INDEX.get(tet)
            if idx is None:
                raise ValueError(f"not an unambiguous tetranucleotide: {tet!r}")
            z[i, idx] = val
            present[i, idx] = 1
    try:
        m = eng.tetra_corr(z, present) if n > 1 else np.ones((n, n))
    except _lib.PyaniGpuError as exc:
        if exc.code == _lib.PG_E_KEYSET:
            raise AssertionError() from exc
        if exc.code == _lib.PG_E_EMPTY:
            raise ZeroDivisionError("division by zero") from exc
        raise
    return pd.DataFrame(m, index=orgs, columns=orgs, dtype=float)


def calculate_tetra(infiles: Iterable, engine: Engine = None) -> pd.DataFrame:
    """Z-scores + correlations for a list of FASTA files in ONE device-resident pass (no host round trip)."""
    eng = engine or default_engine()
    files = [Path(f) for f in infiles]
    stems = [f.stem for f in files]
    order = sorted(range(len(files)), key=lambda k: stems[k])
    scratch_store = eng.genome_count() == 0
    labels = [stems[k] for k in order]
    try:
        ids = [g for g, _, _ in eng.add_fasta_batch([files[k] for k in order])]
        _, _, corr = eng.tetra_matrix(ids, want_corr=True)
    except _lib.PyaniGpuError as exc:
        if exc.code == _lib.PG_E_KEYSET:
            raise AssertionError() from exc
        if exc.code == _lib.PG_E_EMPTY:
            raise ZeroDivisionError("division by zero") from exc
        raise
    finally:
        if scratch_store:       # on the error paths too
            eng.clear_genomes()
    return pd.DataFrame(corr, index=labels, columns=labels, dtype=float)


def write_correlations_tab(df: pd.DataFrame, path) -> None:
    """TETRA_correlations.tab in the reference's format (average_nucleotide_identity.py:782-787)."""
    df.to_csv(path, index=True, sep="\t")
