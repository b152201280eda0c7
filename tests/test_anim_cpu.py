"""CPU-only: the ANIm reduction oracle is pinned to the reference's known answers, and the host build of the engine's
1-to-1 filter (the same pga::lis_filter source the HIP kernels compile) reproduces real `delta-filter -1` output."""
import csv
import json
import subprocess
from pathlib import Path

import pytest

from tests.conftest import GOLD, ROOT

import sys
sys.path.insert(0, str(ROOT / "oracle"))
import anim_oracle  # noqa: E402


def test_parse_delta_known_answers():
    """tests/test_anim.py:96-100 and tests/test_parsing.py:52-64 of the reference."""
    assert anim_oracle.parse_delta(GOLD / "anim" / "test.delta.gz") == (4016947, 4017751, 0.9994621994447228, 2191)


def test_deltadir_identity_matrix_6dp():
    """tests/fixtures/anim/dataframes/deltadir_result.csv (reference tests/test_anim.py:243-255)."""
    rows = list(csv.reader(open(GOLD / "ref_targets" / "anim_deltadir_result.csv")))
    names = rows[0][1:]
    for r in rows[1:]:
        for s, v in zip(names, r[1:]):
            if r[0] != s:
                t = anim_oracle.parse_delta(GOLD / "anim" / "caulobacter" / f"{r[0]}_vs_{s}.filter.gz")
                assert f"{t[2]:.6f}" == v


def test_goldens_file_consistent():
    gold = json.loads((GOLD / "anim_goldens.json").read_text())
    assert gold["reference_known_answers"]["test.delta"] == gold["parse_delta"]["test.delta"]
    for rel, tup in gold["parse_delta"].items():
        assert list(anim_oracle.parse_delta(GOLD / "anim" / (rel + ".gz"))) == tup


def test_legacy_matrix_assembly_overwrite_order():
    """process_deltadir semantics (anim.py:487-496, pyani_tools.py:108-167): B_vs_A overwrites the mirrored cells."""
    res = {("A", "B"): (90, 80, 0.9, 5), ("B", "A"): (70, 60, 0.8, 7)}
    m = anim_oracle.anim_matrices(res, {"A": 100, "B": 200})
    assert m["alignment_lengths"]["A"]["B"] == 60.0 and m["alignment_lengths"]["B"]["A"] == 70.0
    assert m["percentage_identity"]["A"]["B"] == 0.9 and m["percentage_identity"]["B"]["A"] == 0.8
    assert m["similarity_errors"]["A"]["B"] == 7.0
    assert m["alignment_coverage"]["A"]["B"] == 60 / 100 and m["alignment_coverage"]["B"]["A"] == 70 / 200
    assert m["alignment_lengths"]["A"]["A"] == 100.0


@pytest.fixture(scope="module")
def filter_check():
    exe = ROOT / "tools" / "anim_debug" / "filter_check"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT / 'pyani_amd' / 'csrc'}", str(exe) + ".cpp", "-o", str(exe)], check=True)
    return exe


def test_one_to_one_filter_matches_delta_filter(filter_check):
    """27 real .delta -> .filter pairs (MUMmer 3.1/3.23 output held by the reference's tests, 12 734 alignments):
    the engine's filter reproduces every keep/drop decision of delta-filter -1."""
    total = wrong = exact_files = 0
    files = sorted((GOLD / "anim").glob("*/*.delta.gz"))
    assert len(files) == 27
    for f in files:
        al, _, _ = anim_oracle.read_delta(f)
        fl, _, _ = anim_oracle.read_delta(str(f).replace(".delta.gz", ".filter.gz"))
        inp = "".join(f"{a.ref_id} {a.qry_id} {a.rs} {a.re} {a.qs} {a.qe} {a.errors}\n" for a in al)
        out = subprocess.run([str(filter_check)], input=inp, capture_output=True, text=True, check=True).stdout
        mine = {tuple(int(x) for x in ln.split()) for ln in out.splitlines()}
        want = {(a.rs, a.re, a.qs, a.qe, a.errors) for a in fl}
        total += len(al)
        wrong += len(mine ^ want)
        exact_files += mine == want
    assert total == 12734 and wrong == 0, (wrong, total)
    assert exact_files == 27
