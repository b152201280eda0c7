"""oracle/blastn_oracle.cpp — the INDEPENDENT restatement of blastn for pyani's ANIb command line (pyani/anib.py:451-471) — pinned on
the BLAST+ tables the reference's own tests hold (tests/fixtures/anib/blastn/*.blast_tab -> tests/golden/anib/, copied as data).

BLAST+ is third-party and absent (it can be neither built nor run here), so its published algorithm is restated; what pins the
restatement are these tables, ROW BY ROW: (fragment, length, mismatch, gaps, qstart, qend, sstart, send, subject record).  Measured on
all 12 tables (tools/blastn_oracle_agreement.py -> profiles/r06_blastn_oracle_vs_blastplus.json): 99.86 - 99.95 % of BLAST+'s rows
reproduced exactly, 99.85 - 100 % of the rows parse_blast_tab uses.  One table (NC_002696_vs_NC_010338) was used to settle two
constants of the traceback start rule; the other eleven are held out.  The CPU suite runs slices (seconds); the full tables are the tool."""
import re
import sys

import pytest

from tests.conftest import GOLD, ROOT

sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def agreement():
    import blastn_oracle_agreement as A
    return A


@pytest.mark.parametrize("pair,frags", [("NC_010338_vs_NC_014100", 400), ("NC_014100_vs_NC_002696", 400), ("NC_011916_vs_NC_010338", 400)])
def test_oracle_reproduces_blast_plus_rows_on_held_out_tables(agreement, pair, frags):
    """The first 400 fragments of three HELD-OUT tables (78 - 84 % identity; the second has a two-record subject, so -max_target_seqs 1
    is exercised): >= 99.5 % of BLAST+'s rows identical, >= 99 % of the rows parse_blast_tab uses, its tuple within 0.005 percentage
    points of identity and 0.05 % of aligned length / similarity errors."""
    q, s = pair.split("_vs_")
    rep, blast, ours = agreement.compare(q, s, frags)
    assert rep["rows_blast"] > 600 and rep["used_rows_blast"] > 150, rep
    assert rep["identical_fraction_of_blast"] >= 0.995, rep
    assert rep["used_identical_fraction_of_blast"] >= 0.99 and rep["used_only_blast"] + rep["used_only_oracle"] <= 1, rep
    assert abs(rep["identity_pp_diff"]) <= 0.005 and abs(rep["aln_length_rel_diff"]) <= 5e-4 and abs(rep["sim_errors_rel_diff"]) <= 5e-4, rep


def test_oracle_gives_the_reference_known_answer_on_the_near_identical_pair(agreement):
    """NC_002696 against NC_011916 (99.99 %): every row parse_blast_tab uses equals BLAST+'s, hence the reference's own known answer
    (tests/test_anib.py:387-391: 4 016 551, 93, 99.997 693 577 050 029) comes out of the oracle's table."""
    rep, blast, ours = agreement.compare("NC_002696", "NC_011916")
    assert rep["used_rows_identical"] == rep["used_rows_blast"] == rep["used_rows_oracle"] == 3939, rep
    aln, err, pid = rep["parse_blast_tab_oracle"]
    assert (aln, err) == (4016551, 93) and abs(pid - 99.997693577050029) < 1e-9, rep
    assert rep["identical_fraction_of_blast"] >= 0.998, rep


def test_oracle_shares_no_code_with_the_product():
    """the ANIb search was only ever compared with the product's own header compiled for the host.  This oracle includes
    nothing from pyani_amd/ and the product never loads anything under oracle/."""
    src = (ROOT / "oracle" / "blastn_oracle.cpp").read_text()
    assert not re.search(r'#include\s+"', src), "the oracle must only include standard headers"
    assert "pyani_amd" not in re.sub(r"//.*", "", src)
    for f in (ROOT / "pyani_amd").rglob("*"):
        if f.suffix in (".py", ".h", ".hip", ".inc", ".cpp"):
            assert "blastn_oracle" not in f.read_text(errors="ignore"), f
