"""GPU (through the C ABI) vs the independent nucmer oracle (oracle/nucmer_oracle.cpp, run here on the same FASTA files) on genome
pairs whose RECORDS share content and are cut at different places (tests/fuzz_genomes.py): every alignment record and every indel
list, --mum and --maxmatch.  What it pins: `mummer -mum` tests query-side uniqueness per query SEQUENCE —
a repeat that sits once in each of two contigs is unique in either — and postnuc cuts a cluster that mgaps built across the
junction of two reference records.  pyani/anim.py:240-289 runs nucmer on whatever multi-FASTA files it is given (draft assemblies)."""
import random
import subprocess

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _oracle():
    exe = ROOT / "oracle" / "_build" / "nucmer_oracle"
    exe.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "oracle" / "nucmer_oracle.cpp"), "-o", str(exe)], check=True)
    return exe


def _oracle_records(exe, pa, pb, maxmatch):
    out = subprocess.run([str(exe), str(pa), str(pb), "--delta"] + (["--maxmatch"] if maxmatch else []), capture_output=True, text=True, check=True).stdout
    want, cur = {}, None
    for line in out.splitlines():
        t = line.split()
        if t and t[0] == "ALN":
            cur = (t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]))
            want[cur] = []
        elif cur is not None and len(t) == 1 and t[0] != "0":
            want[cur].append(int(t[0]))
    return want


def test_multirecord_pairs_every_record_and_indel_list_equal_the_nucmer_oracle(tmp_path):
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    from tests.fuzz_genomes import make_pair, write_fasta
    exe = _oracle()
    trials = []
    for t in range(40):
        rng = random.Random(1000003 + t)
        ref, qry = make_pair(rng, 5 if t % 3 == 0 else 4)
        pa, pb = tmp_path / f"r{t}.fna", tmp_path / f"q{t}.fna"
        write_fasta(pa, f"r{t}_", ref)
        write_fasta(pb, f"q{t}_", qry)
        trials.append((pa, pb))
    n_records = 0
    with Engine(0) as eng:
        ids = [(eng.add_fasta(pa)[0], eng.add_fasta(pb)[0]) for pa, pb in trials]
        names = [(anim.fasta_records(pa), anim.fasta_records(pb)) for pa, pb in trials]
        for mm in (False, True):
            sel = [k for k in range(len(trials)) if not mm or k % 4 == 1]
            # both directions of every pair in ONE call: the pair and its reverse share their seeding (mirrored units)
            q = [ids[k][0] for k in sel] + [ids[k][1] for k in sel]
            s = [ids[k][1] for k in sel] + [ids[k][0] for k in sel]
            off, recs, ioff, ind = eng.anim_alignments_batch(q, s, maxmatch=mm, with_indels=True)
            for j, k in enumerate(sel + sel):
                fwd = j < len(sel)
                pa, pb = trials[k] if fwd else trials[k][::-1]
                na, nb = names[k] if fwd else names[k][::-1]
                want = _oracle_records(exe, pa, pb, mm)
                got = {}
                for x in range(int(off[j]), int(off[j + 1])):
                    r = recs[x]
                    key = (na[int(r["ref_rec"])][0], nb[int(r["qry_rec"])][0], int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]), int(r["errors"]))
                    got[key] = [int(v) for v in ind[int(ioff[x]):int(ioff[x + 1])]]
                assert got == want, (mm, k, fwd, len(got), len(want), sorted(set(got) ^ set(want))[:3])
                n_records += len(want)
    assert n_records > 400


def test_multirecord_tuples_without_the_traceback_equal_the_oracle_reduction(tmp_path):
    """The same pairs through pg_anim_pairs (pre-passes on, no traceback): the parse_delta tuple of the unfiltered records equals the
    reduction (oracle/anim_oracle.py) of the nucmer oracle's records."""
    import sys
    sys.path.insert(0, str(ROOT / "oracle"))
    import anim_oracle
    from pyani_amd.engine import Engine
    from tests.fuzz_genomes import make_pair, write_fasta
    exe = _oracle()
    with Engine(0) as eng:
        q, s, want = [], [], []
        for t in range(24):
            rng = random.Random(2000003 + t)
            ref, qry = make_pair(rng, 4)
            pa, pb = tmp_path / f"r{t}.fna", tmp_path / f"q{t}.fna"
            write_fasta(pa, f"r{t}_", ref)
            write_fasta(pb, f"q{t}_", qry)
            q.append(eng.add_fasta(pa)[0])
            s.append(eng.add_fasta(pb)[0])
            recs = [anim_oracle.Aln(k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[6], 0, ()) for k in _oracle_records(exe, pa, pb, False)]
            want.append(anim_oracle.parse_delta_records(recs) if recs else None)
        res = eng.anim_pairs(q, s, filter_1to1=False)
        for k, w in enumerate(want):
            r = res[k]
            if w is None:
                assert int(r["n_alignments"]) == 0
                continue
            assert (int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]), int(r["sim_errors"])) == w, k
