"""ANIm parity on the GPU against the MUMmer output recovered in round 2 (26 nucmer runs, 24 857 alignment records):

  caulobacter  12 ordered pairs of the 4 Caulobacter genomes with real nucmer .delta + delta-filter .filter files
               (tests/fixtures/anim/deltadir); NC_010338 / NC_014100 were recovered from the reference's JSpecies BLAST
               databases (tools/make_goldens.py), so 10 of the 12 pairs (85-87 % identity, ~1200 alignments each) are new
  group2       2 ordered pairs of draft genomes (tests/test_JSpecies/Group_2), raw .delta only
  jspecies     12 raw .delta files JSpecies' own nucmer runs left behind (single-record NC_002696)

With the default extender (MUMmer's own postnuc algorithm, pga_postnuc.inc) every one of those records is reproduced —
coordinates and error counts — and so is every delta-filter decision and every parse_delta tuple, bit for bit.  The per-pair
report goes to gpurun_out/anim_oos_gpu_report.json (committed copy under profiles/).  History: the banded64 extender reached
95.5 % of these records unfitted and 99.55 % after five fitted rules (profiles/archive/r02_anim_oos_*.json); Group_2 and the JSpecies
runs were never used for any choice.
"""
import json
from pathlib import Path

import pytest

from tests.conftest import GOLD, ROOT

import sys
sys.path.insert(0, str(ROOT / "oracle"))
import anim_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def _key(x):
    return (x.ref_id, x.qry_id, x.rs, x.re, x.qs, x.qe, x.errors)


@pytest.fixture(scope="module")
def oos(genome_dir, tmp_path_factory):
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    gold = json.loads((GOLD / "anim_goldens.json").read_text())["parse_delta"]
    tmp = tmp_path_factory.mktemp("oos")
    paths = dict(genome_dir["caulobacter"])
    paths.update(genome_dir["group2"])
    # JSpecies ran nucmer on the single-record NC_002696 (= the fixture file's two records joined)
    body = "".join(l.strip() for l in open(paths["NC_002696"]) if not l.startswith(">"))
    joined = tmp / "NC_002696.fna"
    joined.write_text(">gi|16124256|ref|NC_002696.2| joined\n" + "\n".join(body[i:i + 70] for i in range(0, len(body), 70)) + "\n")
    report = {}
    with Engine(0) as eng:
        added = {s: eng.add_fasta(p) for s, p in paths.items()}
        added["NC_002696@joined"] = eng.add_fasta(joined)
        ids = {s: v[0] for s, v in added.items()}
        glen = {s: int(v[1]) for s, v in added.items()}          # genome length = sum of the record lengths (pyani_files.py:128-142)
        recs = {s: anim.fasta_records(p) for s, p in paths.items()}
        recs["NC_002696@joined"] = anim.fasta_records(joined)
        for grp in ("caulobacter", "group2", "jspecies"):
            for f in sorted((GOLD / "anim" / grp).glob("*.delta.gz")):
                a, b = f.name[:-len(".delta.gz")].split("_vs_")
                ka, kb = (s + "@joined" if grp == "jspecies" and s == "NC_002696" else s for s in (a, b))
                al = eng.anim_pair_alignments(ids[ka], ids[kb])
                mine = {(recs[ka][int(x["ref_rec"])][0], recs[kb][int(x["qry_rec"])][0], int(x["rs"]), int(x["re"]), int(x["qs"]),
                         int(x["qe"]), int(x["errors"])): int(x["kept"]) for x in al}
                want = {_key(x) for x in anim_oracle.read_delta(f)[0]}
                rep = {"mummer_records": len(want), "ours": len(mine), "exact": len(want & set(mine))}
                allr = [anim_oracle.Aln(*k, k[6], 0, ()) for k in mine]
                kept = [anim_oracle.Aln(*k, k[6], 0, ()) for k, v in mine.items() if v == 3]
                cmp = [("delta", gold[f"{grp}/{a}_vs_{b}.delta"], allr)]
                flt = Path(str(f).replace(".delta.gz", ".filter.gz"))
                if flt.exists():
                    want_f = {_key(x) for x in anim_oracle.read_delta(flt)[0]}
                    rep["filter_records"], rep["filter_exact"] = len(want_f), len(want_f & {_key(x) for x in kept})
                    cmp.append(("filter", gold[f"{grp}/{a}_vs_{b}.filter"], kept))
                for name, m, rs in cmp:
                    o = anim_oracle.parse_delta_records(rs)
                    rep[name] = {"mummer": m, "ours": list(o), "identity_abs_diff": abs(m[2] - o[2]),
                                 "ref_aln_len_rel_diff": abs(m[0] - o[0]) / m[0], "qry_aln_len_rel_diff": abs(m[1] - o[1]) / m[1],
                                 # coverage as pyani reports it: aligned length / genome length (anim.py:470-480)
                                 "ref_coverage_abs_diff": abs(m[0] - o[0]) / glen[ka], "qry_coverage_abs_diff": abs(m[1] - o[1]) / glen[kb]}
                report[f"{grp}/{a}_vs_{b}"] = rep
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    tot = sum(r["mummer_records"] for r in report.values())
    ex = sum(r["exact"] for r in report.values())
    (out / "anim_oos_gpu_report.json").write_text(json.dumps({"total_records": tot, "exact": ex, "pairs": report}, indent=1, sort_keys=True))
    return report


def test_every_mummer_record_is_reproduced(oos):
    """26 runs: every alignment record of nucmer's .delta (coordinates + error count) and nothing else; where a .filter file
    exists, exactly delta-filter -1's records are flagged kept."""
    assert len(oos) == 26
    for name, r in oos.items():
        assert r["exact"] == r["mummer_records"] == r["ours"], (name, r["exact"], r["mummer_records"], r["ours"])
        if "filter_records" in r:
            assert r["filter_exact"] == r["filter_records"], (name, r["filter_exact"], r["filter_records"])
    assert sum(r["mummer_records"] for r in oos.values()) == 24857


def test_parse_delta_tuples_are_identical(oos):
    """pyani's parse_delta over the engine's records == over MUMmer's, filtered and unfiltered: aligned lengths, error counts
    and identity to the last bit (BASELINE.json asks for 1e-4)."""
    n = 0
    for name, r in oos.items():
        for k in ("delta", "filter"):
            if k in r:
                assert r[k]["ours"] == r[k]["mummer"], (name, k, r[k])
                n += 1
    assert n == 26 + 12
