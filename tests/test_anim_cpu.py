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
    files = [f for f in sorted((GOLD / "anim").glob("*/*.delta.gz")) if Path(str(f).replace(".delta.gz", ".filter.gz")).exists()]
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


def test_python_restatement_of_the_one_to_one_filter_matches_delta_filter():
    """oracle/anim_oracle.py:delta_filter_1to1 — the INDEPENDENT (pure-Python) restatement of `delta-filter -1`, the one the nucmer
    oracle's records go through when whole parse_delta tuples are compared (bench.py's CPU leg, tests) — on the same 27 real pairs."""
    files = [f for f in sorted((GOLD / "anim").glob("*/*.delta.gz")) if Path(str(f).replace(".delta.gz", ".filter.gz")).exists()]
    checked = 0
    for f in files:
        al, _, _ = anim_oracle.read_delta(f)
        fl, _, _ = anim_oracle.read_delta(str(f).replace(".delta.gz", ".filter.gz"))
        keep = anim_oracle.delta_filter_1to1(al)
        got = sorted((a.ref_id, a.qry_id, a.rs, a.re, a.qs, a.qe, a.errors) for a, k in zip(al, keep) if k)
        assert got == sorted((a.ref_id, a.qry_id, a.rs, a.re, a.qs, a.qe, a.errors) for a in fl), f.name
        checked += 1
    assert checked == 27


def test_run_matrices_vectorised_equals_cellwise_definition():
    """assemble_run_matrices (pyani_orm.update_comparison_matrices semantics, vectorised) against the cell-by-cell
    definition: [q, s] cells only, diagonals 1 / 1 / length / 0 / 1, hadamard = identity * cov_query."""
    import numpy as np
    from pyani_amd import anim
    rng = np.random.default_rng(7)
    labels = [f"g{k:02d}" for k in rng.permutation(12)]
    lengths = {g: int(rng.integers(700_000, 6_000_000)) for g in labels}
    res = {}
    for q in labels:
        for s in labels:
            if q != s and rng.random() < 0.8:          # some pairs have no result (no alignment)
                res[(q, s)] = (int(rng.integers(1, lengths[q])), int(rng.integers(1, lengths[s])), float(rng.random()),
                               int(rng.integers(0, 50_000)))
    m = anim.assemble_run_matrices(res, lengths)
    order = sorted(labels)
    for name in ("identity", "coverage", "aln_lengths", "sim_errors", "hadamard"):
        assert list(m[name].index) == order and list(m[name].columns) == order
    for q in order:
        for s in order:
            if q == s:
                want = (1.0, 1.0, float(lengths[q]), 0.0, 1.0)
            elif (q, s) in res:
                qa, _, pid, err = res[(q, s)]
                want = (pid, qa / lengths[q], float(qa), float(err), pid * (qa / lengths[q]))
            else:
                want = (np.nan,) * 5        # no comparison: the reference's frames start as NaN (pyani_orm.py:627-637)
            got = tuple(float(m[n].loc[q, s]) for n in ("identity", "coverage", "aln_lengths", "sim_errors", "hadamard"))
            assert got == want or (all(np.isnan(got)) and all(np.isnan(want))), (q, s)
    # the form the reference stores: integer genome_id index (sorted), DataFrame.to_json() strings, Comparison rows
    gids = {g: 100 - k for k, g in enumerate(order)}                      # ids in reverse label order
    mi = anim.assemble_run_matrices(res, lengths, genome_ids=gids)
    assert list(mi["identity"].index) == sorted(gids.values()) == list(mi["identity"].columns)
    for name in m:
        for q in order:
            for s in order:
                a, b = float(m[name].loc[q, s]), float(mi[name].loc[gids[q], gids[s]])
                assert a == b or (np.isnan(a) and np.isnan(b))
    js = anim.run_matrices_to_json(mi)
    assert sorted(js) == ["df_alnlength", "df_coverage", "df_hadamard", "df_identity", "df_simerrors"]
    import pandas as pd
    from io import StringIO
    back = pd.read_json(StringIO(js["df_identity"]))
    assert js["df_identity"] == mi["identity"].to_json() and back.shape == (12, 12) and "null" in js["df_identity"]
    rows = anim.comparison_rows(res, lengths, gids, maxmatch=True)
    assert len(rows) == len(res)
    (q, s), (qa, sa, pid, err) = next(iter(res.items()))
    assert rows[0] == {"query_id": gids[q], "subject_id": gids[s], "aln_length": qa, "sim_errs": err, "identity": pid,
                       "cov_query": qa / lengths[q], "cov_subject": sa / lengths[s], "program": anim.PROGRAM, "version": anim.VERSION,
                       "fragsize": None, "maxmatch": True, "kmersize": None, "minmatch": None}


def test_write_delta_round_trip_and_grammar(tmp_path):
    """anim.write_delta: MUMmer .delta grammar (path line, NUCMER, '>' blocks, 7-field headers, 0 terminators), record ids
    and lengths from the FASTA files, filtered / unfiltered variants; the oracle's parse_delta reads the result back."""
    import numpy as np
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    ref = tmp_path / "r.fna"
    qry = tmp_path / "q.fna"
    ref.write_text(">r1 first\n" + "ACGT" * 50 + "\n>r2\n" + "A" * 120 + "\n")
    qry.write_text(">q1\n" + "ACGT" * 40 + "\n" + "GG" * 5 + "\n")
    assert anim.fasta_records(ref) == [("r1", 200), ("r2", 120)] and anim.fasta_records(qry) == [("q1", 170)]
    al = np.zeros(3, dtype=Engine.ALN_DTYPE)
    al[0] = (0, 0, 1, 160, 1, 160, 2, 3)
    al[1] = (1, 0, 5, 60, 170, 115, 1, 3)      # reverse strand: qs > qe
    al[2] = (0, 0, 150, 200, 100, 150, 9, 1)   # dropped by the 1-to-1 filter
    out = tmp_path / "r_vs_q.filter"
    assert anim.write_delta(out, ref, qry, al, filtered=True) == 2
    lines = out.read_text().splitlines()
    assert lines[1] == "NUCMER" and lines[0].split()[0].endswith("r.fna")
    assert lines[2] == ">r1 q1 200 170" and lines[3] == "1 160 1 160 2 2 0" and lines[4] == "0"
    assert lines[5] == ">r2 q1 120 170" and lines[6] == "5 60 170 115 1 1 0"
    # reference intervals on two sequences: 160 + 56; query intervals 1-160 and 115-170 on one sequence: union 170
    assert anim_oracle.parse_delta(out) == (160 + 56, 170, (160 * 2 - 4 + 56 * 2 - 2) / (160 * 2 + 56 * 2), 3)
    out2 = tmp_path / "r_vs_q.delta"
    assert anim.write_delta(out2, ref, qry, al, filtered=False) == 3
    assert len(anim_oracle.read_delta(out2)[0]) == 3


def test_write_delta_with_indel_lists_reproduces_mummer_files(tmp_path, genome_dir):
    """anim.write_delta(indels=...): fed with the records and indel lists of MUMmer's own files (read back by the oracle's parser),
    it writes those files again line for line — .delta and .filter, single- and multi-record genomes (what the GPU test
    tests/test_anim_delta_gpu.py then checks is only where the records and lists come from)."""
    import gzip
    import numpy as np
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    cases = [("blochmannia", "GCF_000011745.1_ASM1174v1_genomic_vs_GCF_000043285.1_ASM4328v1_genomic", "delta"),
             ("blochmannia", "GCF_000011745.1_ASM1174v1_genomic_vs_GCF_000043285.1_ASM4328v1_genomic", "filter"),
             ("caulobacter", "NC_011916_vs_NC_002696", "delta")]
    for grp, name, ext in cases:
        gold = GOLD / "anim" / grp / f"{name}.{ext}.gz"
        a, b = name.split("_vs_")
        fa, fb = genome_dir[grp][a], genome_dir[grp][b]
        ra = {rid: k for k, (rid, _) in enumerate(anim.fasta_records(fa))}
        rb = {rid: k for k, (rid, _) in enumerate(anim.fasta_records(fb))}
        recs = anim_oracle.read_delta(gold)[0]
        al = np.zeros(len(recs), dtype=Engine.ALN_DTYPE)
        for k, x in enumerate(recs):
            al[k] = (ra[x.ref_id], rb[x.qry_id], x.rs, x.re, x.qs, x.qe, x.errors, 3)
        out = tmp_path / f"{name}.{ext}"
        assert anim.write_delta(out, fa, fb, al, filtered=(ext == "filter"), indels=[list(x.indels) for x in recs]) == len(recs)
        assert out.read_text().splitlines()[1:] == gzip.open(gold, "rt").read().splitlines()[1:], (grp, name, ext)


class OracleEngine:
    """Engine.anim_reduce restated with the oracle's parse_delta arithmetic (CPU stand-in for the host-logic tests; the GPU
    reduction itself is compared bit for bit in tests/test_anim_gpu.py)."""

    def anim_reduce(self, files, apply_filter=False):
        import numpy as np
        from collections import namedtuple
        from pyani_amd import _lib
        from pyani_amd.engine import Engine
        Rec = namedtuple("Rec", "ref_id qry_id rs re qs qe errors")
        out = np.zeros(len(files), dtype=Engine.ANIM_DTYPE)
        for i, recs in enumerate(files):
            try:
                t = anim_oracle.parse_delta_records([Rec(*r) for r in recs])
            except ZeroDivisionError:
                out[i]["status"] = _lib.PG_ANIM_NO_ALIGNMENT
                continue
            out[i]["ref_aln_len"], out[i]["qry_aln_len"], out[i]["identity"], out[i]["sim_errors"] = t
        return out


@pytest.mark.parametrize("case", ["caulobacter_deltadir", "prefix_stems"])
def test_process_deltadir_equals_the_imported_reference(tmp_path, case):
    """All five matrices of pyani.anim.process_deltadir, produced by importing the reference (tools/make_goldens.py), on its
    own deltadir fixture and on a directory whose stems are prefixes of one another (sorted-Path order != sorted-string
    order: ADVICE r01) with missing directions: bit-equal, NaN where the reference has NaN."""
    import gzip
    import json
    import numpy as np
    from pyani_amd import anim
    fx = json.loads((GOLD / "ref_targets" / "anim_process_deltadir_cases.json").read_text())[case]
    fx["lengths"] = dict(fx["lengths"])      # stored as an ordered list: the dict order is the label order
    if "files" in fx:
        for rel, text in fx["files"].items():
            (tmp_path / rel).parent.mkdir(parents=True, exist_ok=True)
            (tmp_path / rel).write_text(text)
    else:
        for gz in sorted((GOLD / "anim" / "caulobacter").glob("*.filter.gz")):
            q = gz.name.split("_vs_")[0]
            (tmp_path / q).mkdir(exist_ok=True)
            with gzip.open(gz, "rb") as fi:
                (tmp_path / q / gz.name[:-3]).write_bytes(fi.read())
    res = anim.process_deltadir(tmp_path, fx["lengths"], engine=OracleEngine())
    got = dict((stem, df) for df, stem in res.data)
    assert sorted(got) == sorted(fx["matrices"])
    for stem, want in fx["matrices"].items():
        df = got[stem]
        assert list(df.index) == want["labels"] == list(df.columns)
        for a, row in zip(want["labels"], want["rows"]):
            for b, h in zip(want["labels"], row):
                w, g = float.fromhex(h), float(df.loc[a, b])
                assert g == w or (np.isnan(g) and np.isnan(w)), (stem, a, b, g, w)
    # the pair-tuple route (what calculate_anim_pairs feeds) gives the same matrices with the default file order
    tuples = {tuple(f.stem.split("_vs_")): anim_oracle.parse_delta(f) for f in sorted(tmp_path.glob("*/*.filter"))}
    again = anim.assemble_legacy_results(tuples, fx["lengths"])
    for (df, stem), (df2, _) in zip(res.data, again.data):
        assert df.equals(df2), stem


def test_duplicate_stems_are_rejected(tmp_path):
    """pyani keys every result by Path.stem; two inputs with one stem would silently overwrite each other."""
    from pyani_amd import anim

    class NoEngine:
        def genome_count(self):
            return 0
    (tmp_path / "a").mkdir(), (tmp_path / "b").mkdir()
    for d in ("a", "b"):
        (tmp_path / d / "same.fna").write_text(">x\nACGT\n")
    with pytest.raises(ValueError, match="share a stem"):
        anim.calculate_anim_pairs([tmp_path / "a" / "same.fna", tmp_path / "b" / "same.fna"], engine=NoEngine())


def test_process_deltadir_walk_and_assembly(tmp_path):
    """pyani.anim.process_deltadir (anim.py:415-497) mirrored by pyani_amd.anim.process_deltadir: the directory walk, the
    skipping of foreign files, the overwrite order and the error behaviour, with the pinned oracle standing in for the
    GPU reduction (the reduction itself is compared on the GPU in tests/test_anim_gpu.py); identities against the
    reference's deltadir_result.csv (tests/test_anim.py:243-255)."""
    import gzip
    import numpy as np
    from pyani_amd import _lib, anim

    from collections import namedtuple
    from pyani_amd.engine import Engine
    Rec = namedtuple("Rec", "ref_id qry_id rs re qs qe errors")

    class OracleEngine:   # Engine.anim_reduce restated with the oracle's parse_delta arithmetic
        def anim_reduce(self, files, apply_filter=False):
            out = np.zeros(len(files), dtype=Engine.ANIM_DTYPE)
            for i, recs in enumerate(files):
                try:
                    t = anim_oracle.parse_delta_records([Rec(*r) for r in recs])
                except ZeroDivisionError:
                    out[i]["status"] = _lib.PG_ANIM_NO_ALIGNMENT
                    continue
                out[i]["ref_aln_len"], out[i]["qry_aln_len"], out[i]["identity"], out[i]["sim_errors"] = t
            return out

    rows = list(csv.reader(open(GOLD / "ref_targets" / "anim_deltadir_result.csv")))
    names = rows[0][1:]
    for q in names:
        (tmp_path / q).mkdir()
        for s in names:
            if q != s:
                with gzip.open(GOLD / "anim" / "caulobacter" / f"{q}_vs_{s}.filter.gz", "rb") as fi:
                    (tmp_path / q / f"{q}_vs_{s}.filter").write_bytes(fi.read())
    (tmp_path / names[0] / f"{names[0]}_vs_stranger.filter").write_text("x y\nNUCMER\n")   # foreign file: skipped
    lengths = {n: 4_000_000 + i for i, n in enumerate(names)}
    res = anim.process_deltadir(tmp_path, lengths, engine=OracleEngine())
    for r in rows[1:]:
        for s, v in zip(names, r[1:]):
            if r[0] != s:
                assert f"{res.percentage_identity.loc[r[0], s]:.6f}" == v
            else:
                assert res.alignment_lengths.loc[s, s] == lengths[s]
    q, s = names[0], names[1]   # the mirrored cells hold what the LATER file (sorted order) wrote
    later = anim_oracle.parse_delta(GOLD / "anim" / "caulobacter" / f"{max(q, s)}_vs_{min(q, s)}.filter.gz")
    assert res.similarity_errors.loc[q, s] == res.similarity_errors.loc[s, q] == later[3]
    with pytest.raises(anim.PyaniANImException):
        anim.process_deltadir(tmp_path / names[0] / "nothing_here", lengths, engine=OracleEngine())
    (tmp_path / q / f"{q}_vs_{s}.filter").write_text("a b\nNUCMER\n")   # a run without alignments
    with pytest.raises(ZeroDivisionError):
        anim.process_deltadir(tmp_path, lengths, engine=OracleEngine())


def test_run_anim_recovery_mode_from_real_mummer_output(tmp_path, genome_dir):
    """pyani_amd.subcmd_anim.run_anim (the subcmd_anim.py:128-305 driver minus the database) in --recovery mode over the
    reference's own deltadir fixture: all 12 comparisons are recovered from MUMmer's .filter files, nothing is recomputed,
    identities equal deltadir_result.csv, rows / matrices / JSON follow the ORM semantics (genome ids in sorted path order)."""
    import gzip
    import numpy as np
    from pyani_amd import files, subcmd_anim

    class RecoveryOnlyEngine(OracleEngine):
        def genome_count(self):
            return 0

        def add_fasta_batch(self, paths):
            lens = files.get_sequence_lengths(paths)
            return [(k, lens[Path(p).stem], 1) for k, p in enumerate(paths)]

        def anim_pairs(self, *a, **k):
            raise AssertionError("recovery mode must not recompute a comparison whose output exists")

        def clear_genomes(self):
            pass

    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    stems = sorted(genome_dir["caulobacter"])
    for s in stems:
        (indir / f"{s}.fna").write_bytes(genome_dir["caulobacter"][s].read_bytes())
    (indir / "notes.txt").write_text("not a FASTA file")
    for gz in sorted((GOLD / "anim" / "caulobacter").glob("*.filter.gz")):
        q, s = gz.name[:-len(".filter.gz")].split("_vs_")
        f = subcmd_anim.outfile_path(outdir, q, s)
        f.parent.mkdir(parents=True, exist_ok=True)
        with gzip.open(gz, "rb") as fi:
            f.write_bytes(fi.read())
    run = subcmd_anim.run_anim(indir, outdir, recovery=True, engine=RecoveryOnlyEngine())
    assert run.genome_ids == {s: k + 1 for k, s in enumerate(stems)} and len(run.recovered) == 12 and not run.written
    rows = list(csv.reader(open(GOLD / "ref_targets" / "anim_deltadir_result.csv")))
    names = rows[0][1:]
    for r in rows[1:]:
        for s, v in zip(names, r[1:]):
            if r[0] != s:
                assert f"{run.matrices['identity'].loc[run.genome_ids[r[0]], run.genome_ids[s]]:.6f}" == v
    assert len(run.comparisons) == 12 and {c["program"] for c in run.comparisons} == {"pyani_amd-anim"}
    c = next(c for c in run.comparisons if (c["query_id"], c["subject_id"]) == (run.genome_ids["NC_002696"], run.genome_ids["NC_011916"]))
    want = anim_oracle.parse_delta(GOLD / "anim" / "caulobacter" / "NC_002696_vs_NC_011916.filter.gz")
    assert (c["aln_length"], c["identity"], c["sim_errs"]) == (want[0], want[2], want[3])
    assert c["cov_query"] == want[0] / run.lengths["NC_002696"] and c["cov_subject"] == want[1] / run.lengths["NC_011916"]
    assert run.json["df_identity"] == run.matrices["identity"].to_json() and list(run.matrices["hadamard"].index) == [1, 2, 3, 4]
    assert not np.isnan(run.matrices["coverage"].values).any()
    # one output file missing and no engine able to recompute it -> the engine IS asked for exactly that pair
    subcmd_anim.outfile_path(outdir, "NC_014100", "NC_002696").unlink()
    with pytest.raises(AssertionError, match="must not recompute"):
        subcmd_anim.run_anim(indir, outdir, recovery=True, engine=RecoveryOnlyEngine())


def test_forced_runs_whose_optimal_path_dips_far_below_zero():
    """DESIGN §4, deviation (3), directed: MUMmer's sw_align scores on plain integers, the engine's packed state words hold 15 bits of
    score above a floor (pg_nucmer_core.h: SCORE_BIAS, -2700 since round 5; -1024 before).  tools/anim_debug/forced_check.cpp runs
    pgn::ScalarEngine and the host emulation of the GPU's diagonal-window engines on rectangles whose optimal path falls to -1 586
    (equal to a plain-integer statement of the forced alignment: exact now, lost with the old floor) and to -3 471 (below the floor on
    every path: the run must fail LOUDLY — engine overflow, PG_E_CAPACITY on the pair — and not return an unreachable word's error
    count).  No walk can produce such a rectangle (a backward search breaks 200 anti-diagonals after its last best cell), so there
    is no genome pair that reaches this through the C ABI: the engines are driven directly."""
    exe = ROOT / "tools" / "anim_debug" / "forced_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", f"-I{ROOT / 'pyani_amd' / 'csrc'}", str(exe) + ".cpp", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.count(" ok") == 6 and "WRONG" not in out.stdout, out.stdout


def test_normalised_score_frame_of_the_trimmed_searches_equals_the_plain_one(tmp_path):
    """the wave engines run MUMmer's trimmed searches in a NORMALISED frame (pg_nucmer_diag.h diag_lane_step<NORM>:
    the score field holds score - 3 floor(d / 2) + 32 767, a match keeps the word, the best / threshold words slide with the frame);
    pgn::ScalarEngine's plain words are the definition.  tools/anim_debug/norm_check.cpp drives both directly on searches built for the
    corners of the frame's range — a 10 000 x 10 000 search whose score creeps while the offset runs to 30 000 (live words in the
    bottom of the field), 10 000 matching bases (the top), indels on odd and even anti-diagonals, unclean bases, a band trimmed away, tiny
    and one-row rectangles — forwards and backwards, to the target and to the best cell: end coordinates, errors, score and `reached`
    must be equal in all 100.  A SABOTAGED frame (the offset rising by 2 instead of GOOD_SCORE = 3 per pair of anti-diagonals) must
    be caught: the check is not vacuous."""
    csrc = ROOT / "pyani_amd" / "csrc"
    exe = ROOT / "tools" / "anim_debug" / "norm_check"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{csrc}", str(exe) + ".cpp", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "100 searches on both engines (100 on the emulated wave engine), 0 differ" in out.stdout and "WRONG" not in out.stdout, out.stdout
    # the same program against a copy of the headers whose offset is wrong
    bad = tmp_path / "csrc"
    bad.mkdir()
    for f in csrc.glob("*.h"):
        text = f.read_text()
        if f.name == "pg_nucmer_core.h":
            assert "return GOOD_SCORE * (d >> 1);" in text
            text = text.replace("return GOOD_SCORE * (d >> 1);", "return 2 * (d >> 1);")
        (bad / f.name).write_text(text)
    bexe = tmp_path / "norm_check_sabotaged"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{bad}", str(exe) + ".cpp", "-o", str(bexe)], check=True)
    out = subprocess.run([str(bexe)], capture_output=True, text=True)
    assert out.returncode == 1 and "WRONG" in out.stdout, out.stdout[-2000:]
