"""Full-size ANIm on the GPU (BASELINE.json configs[2] / [3] code path): one whole C3 family — 25 synthetic 5 Mb genomes of
one ancestor, all 600 related ordered pairs — plus 200 unrelated ordered pairs, through pg_anim_pairs in ONE call, against
the CPU statement of the search on an 8-genome subset (oracle/anim_cpu.cpp; fixture made by
tools/make_anim_c3_family_host.py), the size-independent properties of the domain on all of them, a pinned result hash; and the N > 1 path of bench.py on hardware (2 ranks on GPU 0 over gloo)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import GOLD, ROOT

pytestmark = pytest.mark.gpu


def test_c3_family_equals_cpu_statement_and_properties():
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    fx = json.loads((GOLD / "anim_c3_family_host.json").read_text())
    n, L, seed = fx["n"], fx["length"], fx["seed"]
    pairs = [(p[0], p[1]) for p in fx["pairs"]]
    used = sorted({g for p in pairs for g in p})
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*synth.genome(seed, n, g, L)) for g in used}
        eng.upload()
        res = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs])
        lens = {g: eng.genome_length(ids[g])[0] for g in used}
    got = [[int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]).hex(), int(r["sim_errors"]), int(r["n_alignments"]),
            int(r["status"])] for r in res]
    # tuple for tuple against the CPU statement where it was computed (an 8-genome subset of the family: 56 related pairs,
    # + 24 unrelated ones)
    checked = [(p[:2], g, p[2]) for p, g in zip(fx["pairs"], got) if p[2] is not None]
    bad = [c for c in checked if c[1] != c[2]]
    assert len(checked) >= 80 and not bad, f"{len(bad)} of {len(checked)} pairs differ from the CPU statement, first: {bad[:1]}"
    # regression pin of the whole call (the GPU's own results of round 2; changes when the search rules change)
    sha = hashlib.sha1(json.dumps(got).encode()).hexdigest()
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "anim_c3_family_sha1.txt").write_text(sha + "\n")
    pin = GOLD / "anim_c3_family_gpu_sha1.txt"
    if pin.exists():
        assert sha == pin.read_text().strip()
    # size-independent properties (they hold for nucmer + delta-filter -1 + parse_delta output on any input)
    by = {(a, b): r for (a, b), r in zip(pairs, res)}
    n_rel = fx["n_related"]
    for k, ((a, b), r) in enumerate(zip(pairs, res)):
        if k >= n_rel:
            assert int(r["status"]) == 1 and int(r["n_alignments"]) == 0, (a, b)     # unrelated: empty .filter
            continue
        assert int(r["status"]) in (0, 1)
        if int(r["status"]) == 0:
            assert 0 < int(r["ref_aln_len"]) <= lens[a] and 0 < int(r["qry_aln_len"]) <= lens[b]   # interval unions
            assert 0.5 < float(r["identity"]) <= 1.0
            back = by[(b, a)]
            if int(back["status"]) == 0:
                assert abs(float(r["identity"]) - float(back["identity"])) < 0.01, (a, b)
                assert abs(int(r["ref_aln_len"]) - int(back["qry_aln_len"])) < 0.05 * lens[a], (a, b)
    ok = sum(int(r["status"]) == 0 for r in res[:n_rel])
    assert ok >= 0.97 * n_rel


def _bench(extra_env, nproc, tmp_path, tag, args=None):
    env = dict(os.environ, **extra_env)
    args = args or ["--genomes", "24", "--length", "300000", "--seed", "11", "--rows-per-step", "8", "--steps", "3", "--warmup", "0",
                    "--no-cpu-baseline", "--no-tetra"]
    if nproc == 1:
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", "29517", str(ROOT / "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    (tmp_path / f"{tag}.json").write_text(line)
    return json.loads(line)


def test_enqueued_calls_overlap_and_equal_the_blocking_call():
    """pg_anim_pairs_enqueue / _fetch (round 6): the pairs of a C3 family cut into four calls, two in flight at a
    time on disjoint worker slots, give exactly the records of ONE blocking pg_anim_pairs; a third enqueue while two are in flight
    is refused (PG_E_CAPACITY), so is a blocking call; fetching an unknown ticket is an error; an unfetched call does not leak
    (pg_destroy joins it)."""
    from pyani_amd import synth
    from pyani_amd._lib import PyaniGpuError
    from pyani_amd.engine import Engine
    fx = json.loads((GOLD / "anim_c3_family_host.json").read_text())
    n, L, seed = fx["n"], fx["length"], fx["seed"]
    pairs = [(p[0], p[1]) for p in fx["pairs"]][:240]
    used = sorted({g for p in pairs for g in p})
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*synth.genome(seed, n, g, L)) for g in used}
        r, q = [ids[a] for a, _ in pairs], [ids[b] for _, b in pairs]
        want = eng.anim_pairs(r, q)
        cuts = [0, 60, 120, 180, 240]
        got = np.zeros(len(pairs), dtype=want.dtype)
        pending = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            pending.append((lo, hi, eng.anim_pairs_enqueue(r[lo:hi], q[lo:hi])))
            if len(pending) == 2:
                with pytest.raises(PyaniGpuError):
                    eng.anim_pairs_enqueue(r[:4], q[:4])          # two in flight: refused
                with pytest.raises(PyaniGpuError):
                    eng.anim_pairs(r[:4], q[:4])                  # the blocking call owns every worker slot
                a, b, t = pending.pop(0)
                got[a:b] = eng.anim_pairs_fetch(t)
        for a, b, t in pending:
            got[a:b] = eng.anim_pairs_fetch(t)
        assert got.tobytes() == want.tobytes()
        with pytest.raises(PyaniGpuError):
            eng.anim_pairs_fetch((12345, 4))
        assert eng.anim_pairs(r[:8], q[:8]).tobytes() == want[:8].tobytes()      # the blocking call works again
        eng.anim_pairs_enqueue(r[:8], q[:8])                                    # never fetched: the context's teardown waits for it


def test_bench_two_ranks_on_one_gpu_equal_single_rank(tmp_path):
    """bench.py's N > 1 path (rows dealt over the ranks, Engine per rank, one all-gather per step) on hardware: two ranks
    sharing GPU 0 over gloo (PYANI_BENCH_DEBUG_ONE_GPU) give the same full-grid result hash as one rank."""
    one = _bench({}, 1, tmp_path, "one")
    two = _bench({"PYANI_BENCH_DEBUG_ONE_GPU": "1"}, 2, tmp_path, "two")
    assert one["config"]["results_sha1_full_grid"] and one["config"]["results_sha1_full_grid"] == two["config"]["results_sha1_full_grid"]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and one["config"]["pairs_timed"] == two["config"]["pairs_timed"] == 24 * 23
    assert one["config"]["related_pairs_with_alignment"] == two["config"]["related_pairs_with_alignment"] > 0
    for rec in (one, two):
        assert rec["roofline"]["kernel"].startswith("anim_") and rec["roofline"]["achieved"] > 0


def test_bench_four_ranks_on_one_gpu_equal_single_rank_with_either_dealing(tmp_path):
    """FOUR ranks on GPU 0 over gloo, each with its own Engine (two host workers, its own scratch — sized
    against the HBM that is free at the time: pg_api.cpp anim_match_budget) — the configuration in which host threads, streams and
    scratch of several processes contend for one device.  With the default dealing (round 5: the fixed scrambled deal, one engine
    call per rank and step) and with --dynamic-deal (the step's rows PULLED from the cross-rank counter, RowQueue in the job's
    rendezvous store): the same full-grid hash as one rank."""
    args = ["--genomes", "48", "--length", "300000", "--seed", "11", "--rows-per-step", "24", "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--no-tetra"]
    one = _bench({}, 1, tmp_path, "one48", args)
    four = _bench({"PYANI_BENCH_DEBUG_ONE_GPU": "1"}, 4, tmp_path, "four48", args)
    assert one["config"]["results_sha1_full_grid"] and one["config"]["results_sha1_full_grid"] == four["config"]["results_sha1_full_grid"]
    assert four["n_gpus"] == 4 and four["imbalance"]["chunks_per_rank_last_step"] == [1, 1, 1, 1] and four["imbalance"]["dealing"].startswith("fixed scrambled deal")
    dyn = _bench({"PYANI_BENCH_DEBUG_ONE_GPU": "1"}, 4, tmp_path, "four48d", args + ["--dynamic-deal"])
    assert dyn["config"]["results_sha1_full_grid"] == one["config"]["results_sha1_full_grid"]
    assert "job-store" in dyn["imbalance"]["dealing"] and sum(dyn["imbalance"]["chunks_per_rank_last_step"]) >= 4


def test_bench_fragment_mode_two_ranks_equal_single_rank(tmp_path):
    """The same for `bench.py --workload anib` (fragment mode, mixed-length genomes): per-pair results gathered from two ranks
    equal one rank's (identities, coverage, hit counts of the JSON line)."""
    args = ["--workload", "anib", "--genomes", "12", "--rows-per-step", "6", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]
    one = _bench({}, 1, tmp_path, "anib_one", args)
    two = _bench({"PYANI_BENCH_DEBUG_ONE_GPU": "1"}, 2, tmp_path, "anib_two", args)
    for key in ("pairs_timed", "related_pairs_timed", "fragments_timed", "related_pairs_with_hits", "identity_related_min_med_max",
                "coverage_related_median"):
        assert one["config"][key] == two["config"][key], key
    assert one["config"]["pairs_timed"] == 12 * 11 and one["config"]["related_pairs_with_hits"] == 12 * 11 and two["n_gpus"] == 2


def test_bench_rccl_path_with_one_rank_equals_plain_run(tmp_path):
    """The collective path of bench.py with its real backend: torch.distributed.run with ONE rank and PYANI_BENCH_FORCE_DIST
    (init_process_group("nccl"), all_gather_into_tensor of CUDA tensors, barrier, all_reduce) gives the result hash of the plain
    single-GPU run.  (Two RCCL ranks cannot share one GPU; the two-rank tests above therefore run over gloo.)"""
    one = _bench({}, 1, tmp_path, "plain")
    args = ["--genomes", "24", "--length", "300000", "--seed", "11", "--rows-per-step", "8", "--steps", "3", "--warmup", "0",
            "--no-cpu-baseline", "--no-tetra"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", str(ROOT / "bench.py"), "--gpus", "1"] + args
    r = subprocess.run(cmd, env=dict(os.environ, PYANI_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rccl = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rccl["config"]["results_sha1_full_grid"] == one["config"]["results_sha1_full_grid"]
    assert rccl["config"]["pairs_timed"] == one["config"]["pairs_timed"] == 24 * 23
