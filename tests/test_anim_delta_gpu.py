"""The files pyani's nucmer jobs leave on disk, written from ONE batched call with the GPU traceback pass
(pg_anim_alignments_batch(with_indels=1) -> pyani_amd.anim.write_delta): compared with MUMmer's own .delta / .filter files
(tests/golden/anim) LINE FOR LINE — alignment headers, error counts, indel offset lists, order of the records.

  blochmannia  every fixture pair of the 7 Blochmannia genomes (the in-sample set of rounds 1-2): whole files equal
  caulobacter  two pairs at 85-87 % identity (~1200 alignments each, wide forced runs): every record with its indel list equal;
               the record ORDER equals MUMmer's except where a forward- and a reverse-strand cluster start on the same reference
               base — MUMmer's unstable sort of the clusters decides those (DESIGN.md §5b "Traceback")

pyani/nucmer.py:170-290 reads these files (DeltaAlignment / DeltaComparison); anim.py:292-411 (parse_delta) reduces them.
"""
import gzip
from pathlib import Path

import pytest

from tests.conftest import GOLD

pytestmark = pytest.mark.gpu


def _body(path):
    """lines of a .delta / .filter file without the first one (the two FASTA paths of the run that made it)"""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as fh:
        return fh.read().splitlines()[1:]


def _chunks(lines):
    """{sequence-pair header: [alignment header + its indel lines, ...]} in file order"""
    out, hdr, cur = {}, None, None
    for ln in lines[1:]:            # lines[0] == "NUCMER"
        if ln.startswith(">"):
            hdr = ln
            out[hdr] = []
            cur = None
        elif cur is None:
            cur = [ln]
        else:
            cur.append(ln)
            if ln == "0":
                out[hdr].append(tuple(cur))
                cur = None
    return out


def _write_all(eng, genome_paths, pairs, tmp, filtered):
    from pyani_amd import anim
    ids = {}
    for s in sorted({x for p in pairs for x in p}):
        ids[s] = eng.add_fasta(genome_paths[s])[0]
    off, recs, ioff, ind = eng.anim_alignments_batch([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs], with_indels=True)
    files = {}
    for k, (a, b) in enumerate(pairs):
        lo, hi = int(off[k]), int(off[k + 1])
        f = tmp / f"{a}_vs_{b}.{'filter' if filtered else 'delta'}"
        anim.write_delta(f, genome_paths[a], genome_paths[b], recs[lo:hi], filtered=filtered,
                         indels=[ind[int(ioff[x]):int(ioff[x + 1])] for x in range(lo, hi)])
        files[(a, b)] = f
    return files


def test_blochmannia_delta_and_filter_files_equal_mummers_line_for_line(genome_dir, tmp_path):
    from pyani_amd.engine import Engine
    paths = genome_dir["blochmannia"]
    pairs = []
    for f in sorted((GOLD / "anim" / "blochmannia").glob("*.delta.gz")):
        a, b = f.name[:-len(".delta.gz")].split("_vs_")
        if a in paths and b in paths:
            pairs.append((a, b))
    assert len(pairs) >= 15
    with Engine(0) as eng:
        for filtered, ext in ((False, "delta"), (True, "filter")):
            files = _write_all(eng, paths, pairs, tmp_path, filtered)
            eng.clear_genomes()
            checked = 0
            for (a, b), f in files.items():
                gold = GOLD / "anim" / "blochmannia" / f"{a}_vs_{b}.{ext}.gz"
                if not gold.exists():
                    continue
                assert _body(f) == _body(gold), f"{a}_vs_{b}.{ext}"
                checked += 1
            assert checked >= 15


def test_divergent_pairs_every_record_with_its_indel_list(genome_dir, tmp_path):
    from pyani_amd.engine import Engine
    paths = genome_dir["caulobacter"]
    pairs = [("NC_010338", "NC_011916"), ("NC_014100", "NC_010338")]
    with Engine(0) as eng:
        files = _write_all(eng, paths, pairs, tmp_path, False)
    for (a, b), f in files.items():
        mine, gold = _chunks(_body(f)), _chunks(_body(GOLD / "anim" / "caulobacter" / f"{a}_vs_{b}.delta.gz"))
        assert list(mine) == list(gold)                       # the sequence-pair headers, in order
        for hdr in gold:
            assert sorted(mine[hdr]) == sorted(gold[hdr]), hdr     # every record, header line and indel list
            swapped = sum(1 for x, y in zip(mine[hdr], gold[hdr]) if x != y)
            assert swapped <= 8, (hdr, swapped)                 # order: MUMmer's but for same-start clusters of both strands (2 ties here)


def test_batch_records_equal_the_per_pair_call(genome_dir):
    """pg_anim_alignments_batch without the traceback == pg_anim_pair_alignments pair by pair (and == with it, as sets)"""
    from pyani_amd.engine import Engine
    paths = genome_dir["blochmannia"]
    stems = sorted(paths)[:3]
    with Engine(0) as eng:
        ids = [eng.add_fasta(paths[s])[0] for s in stems]
        pairs = [(a, b) for a in ids for b in ids if a != b]
        off, recs, ioff, ind = eng.anim_alignments_batch([a for a, _ in pairs], [b for _, b in pairs])
        assert ioff is None and ind is None
        off2, recs2, _, _ = eng.anim_alignments_batch([a for a, _ in pairs], [b for _, b in pairs], with_indels=True)
        for k, (a, b) in enumerate(pairs):
            one = eng.anim_pair_alignments(a, b)
            got = recs[int(off[k]):int(off[k + 1])]
            assert [tuple(x) for x in got] == [tuple(x) for x in one]
            assert sorted(tuple(x) for x in recs2[int(off2[k]):int(off2[k + 1])]) == sorted(tuple(x) for x in one)


def test_maxmatch_and_mum_lists_on_repeat_genomes_equal_the_nucmer_oracle(tmp_path):
    """Genomes WITH repeats (tests/test_anim_gpu.py::_with_repeats), --mum and --maxmatch: the records AND indel lists of the batched
    call with the traceback pass equal those of oracle/nucmer_oracle.cpp (the restatement of MUMmer's pipeline that reproduces
    every fixture file) run on the same FASTA files — the check behind run_anim(write_output=True, maxmatch=True)."""
    import subprocess
    from pyani_amd import anim, synth
    from pyani_amd.engine import Engine
    from tests.conftest import ROOT
    from tests.test_anim_gpu import _with_repeats
    oracle = ROOT / "oracle" / "_build" / "nucmer_oracle"
    if not oracle.exists():
        oracle.parent.mkdir(exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "oracle" / "nucmer_oracle.cpp"), "-o", str(oracle)], check=True)
    n, L = 4, 300_000
    files = []
    for g in (0, 1):
        seq, off = _with_repeats(*synth.genome(31, n, g, L), g)
        f = tmp_path / f"rep{g}.fna"
        synth.write_fasta(f, seq, off, f"rep{g}")
        files.append(f)
    with Engine(0) as eng:
        ids = [eng.add_fasta(f)[0] for f in files]
        recs_of = [anim.fasta_records(f) for f in files]
        for mm in (False, True):
            off, recs, ioff, ind = eng.anim_alignments_batch([ids[0], ids[1]], [ids[1], ids[0]], maxmatch=mm, with_indels=True)
            for k, (a, b) in enumerate(((0, 1), (1, 0))):
                out = subprocess.run([str(oracle), str(files[a]), str(files[b]), "--delta"] + (["--maxmatch"] if mm else []),
                                     capture_output=True, text=True, check=True).stdout
                want, cur = {}, None
                for line in out.splitlines():
                    t = line.split()
                    if t and t[0] == "ALN":
                        cur = (t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]))
                        want[cur] = []
                    elif cur is not None and len(t) == 1 and t[0] != "0":
                        want[cur].append(int(t[0]))
                got = {}
                for x in range(int(off[k]), int(off[k + 1])):
                    r = recs[x]
                    key = (recs_of[a][int(r["ref_rec"])][0], recs_of[b][int(r["qry_rec"])][0], int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]),
                           int(r["errors"]))
                    got[key] = [int(v) for v in ind[int(ioff[x]):int(ioff[x + 1])]]
                assert got == want, (mm, a, b, len(got), len(want), sorted(set(got) ^ set(want))[:3])
                assert len(got) > 10
