#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: genome-pairs/s for the ANIm N x N grid (BASELINE.json configs[3] = SURVEY.md
§8(d) set C4: 1000 synthetic ~5 Mb genomes, 999 000 ordered pairs), the largest configuration that fits one MI355X and the
one the north-star target is quoted on; with the kernel roofline, the CPU baseline timed on this box's host cores, and the
C2 TETRA roofline nested as a sub-record.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

ANIm (default).  All genomes are resident in HBM on every GPU (2-bit codes + 1-bit mask; 1.9 GB at C4).  A STEP is one
pass of the whole pipeline — seed, cluster, extend, 1-to-1 filter, parse_delta reduction, results back on the host — over
a TILE of the ordered-pair grid: the UNORDERED pairs owned by R genomes (default: a tenth of the genomes per GPU = ~99 900
ordered pairs at C4; `--rows-per-step R`), each in both directions — pyani_amd.parallel.anim_pair_array(symmetric=True): {g, h}
belongs to the smaller id if g + h is even, else to the larger; step k then takes genomes [k*R, (k+1)*R) modulo N).  A pair
and its reverse sit in the same call because they have the same maximal exact matches and the engine seeds them once
(pg_anim.hip "roles").  `value` = ordered pairs processed in the K timed steps / wall time.  N > 1: the rows of every step
are dealt over the ranks (pyani_amd.parallel.anim_row_shard), each rank runs the pairs its rows own, ONE RCCL all-gather per step
(64 B per pair) puts the step's results on every rank.  No other collective, no sequence traffic.  By default a step has
genomes / 10 rows PER GPU (weak scaling: every rank's launches keep their N = 1 size); with --rows-per-step R it has R rows
whatever N is (strong scaling of that step).

  --workload tetra : the TETRA side alone (C2: 200 genomes, counts + Z + Pearson; N > 1 = weak scaling, 200 genomes per GPU).
  --workload anib  : C5 (BASELINE.json configs[4]): 500 genomes of 1-12 Mb, pyani's ANIb on the engine's fragment mode
                     (1020-nt fragments against every other genome); steps of 10 fragmented genomes x 499 subjects.

Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6290 GB/s measured copy


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["anim", "tetra", "anib"], default="anim",
                    help="anim = C4 (default: the N x N ANIm grid the metric is quoted on); tetra = C2 alone; "
                         "anib = C5 (mixed-length set, 1020-nt fragment mode)")
    ap.add_argument("--steps", type=int, default=None, help="default 10 (anim: ten steps of a tenth of the grid = ONE whole C4 grid, so the result hash covers it and no tile is over-represented) / 50 (tetra) / 3 (anib)")
    ap.add_argument("--warmup", type=int, default=None, help="default 2 (anim) / 1 (anib) / 5 (tetra)")
    ap.add_argument("--genomes", type=int, default=None, help="anim: genomes of the job (C4: 1000); tetra: genomes per GPU (C2: 200)")
    ap.add_argument("--length", type=int, default=5_000_000, help="ancestor length in bases (5 Mb)")
    ap.add_argument("--seed", type=int, default=None, help="default: the set's own seed (C4 20250301, C2 20250228)")
    ap.add_argument("--rows-per-step", type=int, default=None, help="genomes (grid rows) per step: default genomes / 10 per GPU (anim: ten steps = the whole grid at N = 1) / 10 (anib)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="anim: ordered pairs timed by the CPU leg (0 = one per host thread, <= 128)")
    ap.add_argument("--cpu-genomes", type=int, default=2, help="tetra: genomes timed by the CPU leg")
    ap.add_argument("--no-tetra", action="store_true", help="anim: skip the nested C2 TETRA sub-record")
    ap.add_argument("--roofline-tiles", type=int, default=0, help="anim: how many tiles the one-worker roofline pass covers (0 = every tile of the grid)")
    ap.add_argument("--side-records", action="store_true", help="anim: keep the side records under --no-cpu-baseline (related_only / unrelated_only / sketch_mode without the CPU leg)")
    ap.add_argument("--no-side-records", action="store_true", help="anim: skip related_only / unrelated_only / strong-step side records")
    ap.add_argument("--dynamic-deal", action="store_true", help="anim, N > 1: the ranks pull a step's rows in guided chunks from a cross-rank counter (round 4's default) "
                                                               "instead of the fixed scrambled deal, one call per rank and step")
    ap.add_argument("--static-deal", action="store_true", help="(the default since round 5; accepted for older command lines)")
    ap.add_argument("--cold-e2e", action="store_true", help="anim, N = 1: measure ONE cold end-to-end run instead of the step loop: FASTA files on disk -> "
                    "parse + pack (pg_add_fasta_batch) -> upload -> seed lists -> the whole grid -> run matrices -> JSON, one wall clock")
    ap.add_argument("--pipeline", action="store_true", help="anim: step k + 1 is enqueued (pg_anim_pairs_enqueue) before step k is fetched, so one step's sequential tail "
                    "overlaps the next step's front.  NOT the default: measured on MI355X at C4's step size (99 900 pairs) it is 3 %% slower than one blocking call per "
                    "step (56 957 vs 58 657 pairs/s, profiles/r06_pipeline_vs_blocking.txt) — the two calls in flight halve each other's launches, and a step this large "
                    "already overlaps its tails across its own two workers; it pays for family-sized calls (related_only.pipelined_calls)")
    ap.add_argument("--expect-sha", default="auto", help="anim: sha1 the whole N x N result grid must have (checked whenever the timed steps cover the grid): "
                    "'auto' = the committed hash of the default C4 job (tests/golden/anim_c4_grid_sha1.txt) when the job IS the default C4 job, 'none' = no check; "
                    "a mismatch fails the run loudly (exit code 3) after the line is printed")
    args = ap.parse_args()
    w = args.workload
    if args.steps is None:
        args.steps = {"anim": 10, "tetra": 50, "anib": 3}[w]
    if args.warmup is None:
        args.warmup = {"anim": 2, "tetra": 5, "anib": 1}[w]
    if args.genomes is None:
        args.genomes = {"anim": 1000, "tetra": 200, "anib": 500}[w]
    if args.seed is None:
        args.seed = {"anim": 20250301, "tetra": 20250228, "anib": 20250302}[w]
    args.rows_default = args.rows_per_step is None
    if args.rows_per_step is None:
        # anim: a tenth of the grid PER GPU per step (a whole C4 grid is an 18 s step; the driver's 25 steps must finish in
        # minutes) — ~100 000 ordered pairs per GPU and step at C4, where launches still fill the GPU and a launch's sequential tails
        # (its longest unit walk, ~0.1 s) stay a fraction of it.  Per-GPU work is fixed as N grows: the
        # default series is WEAK scaling (8 GPUs: 800 rows per step, 10 steps = 8 grids); an explicit --rows-per-step R keeps R rows
        # per step whatever N is (strong scaling of that step).
        world = int(os.environ.get("WORLD_SIZE", "1")) if args.gpus > 1 else 1
        args.rows_per_step = 10 if w == "anib" else max(1, min(args.genomes, (args.genomes // 10) * max(1, world)))
    return args


def synth_genomes(seed, n, L, lo, hi, world):
    from pyani_amd import synth
    with ThreadPoolExecutor(max(1, min(64, (os.cpu_count() or 2) // max(1, world)))) as ex:
        return list(ex.map(lambda g: synth.genome(seed, n, g, L), range(lo, hi)))


# =====================================================================================================================
# ANIm: the N x N grid (C4)
# =====================================================================================================================
ANIM_STAGES = None  # filled from _lib


def anim_cpu_baseline(args, data, n, related, gpu_lookup):
    """The CPU side of the metric on THIS box's host cores, on a bounded stratified sample of the same ordered pairs,
    extrapolated linearly (pairs are independent jobs — pyani's runner is a multiprocessing.Pool of one nucmer +
    delta-filter process per pair, run_multiprocessing.py:130-144).
      kind "reference": nucmer + delta-filter exist on this box -> exactly pyani's jobs, Pool(os.cpu_count()).
      kind "port"     : they do not (MUMmer is third-party, absent from the image) -> the repo's own CPU statement of the
                        same search (oracle/anim_cpu.cpp: the engine's definitions in scalar form on a host k-mer index), one
                        pair per host thread ("own-cpu")."""
    threads = os.cpu_count() or 1
    k = args.cpu_pairs or min(threads, 128)     # bounded: ~2-10 CPU-s per pair, one pair per thread
    K = (n + 24) // 25
    rng = np.random.RandomState(12345)
    rel, unrel = [], []
    while len(rel) < (k + 1) // 2:
        q = int(rng.randint(n)); s = (q + K * int(rng.randint(1, max(2, (n - 1) // K + 1)))) % n
        if s != q and (q % K) == (s % K) and (q, s) not in rel:
            rel.append((q, s))
    while len(unrel) < k // 2:
        q, s = int(rng.randint(n)), int(rng.randint(n))
        if (q % K) != (s % K) and (q, s) not in unrel:
            unrel.append((q, s))
    sample = rel + unrel
    n_rel_job, n_unrel_job = int(related.sum()), int((~related).sum())
    used = sorted({g for p in sample for g in p})
    if shutil.which("nucmer") and shutil.which("delta-filter"):
        try:
            return _nucmer_baseline(sample, len(rel), data, n_rel_job, n_unrel_job, threads)
        except Exception as exc:   # a broken MUMmer install must not cost the GPU line
            note = f"nucmer found but unusable ({exc!r}); "
    else:
        note = "nucmer / delta-filter (MUMmer 3.23) are not installed on this box; "
    sys.path.insert(0, str(ROOT / "oracle"))
    import anim_cpu
    genomes = [data[g] if g in used else None for g in range(n)]
    t0 = time.perf_counter()
    res, secs = anim_cpu.anim_cpu_pairs(genomes, [q for q, _ in sample], [s for _, s in sample], threads=min(threads, len(sample)))
    wall = time.perf_counter() - t0
    t_rel = float(secs[: len(rel)].mean())
    t_unrel = float(secs[len(rel):].mean()) if unrel else 0.0
    job_cpu_s = n_rel_job * t_rel + n_unrel_job * t_unrel
    job_wall = job_cpu_s / threads
    same = covered_n = 0
    for (q, s), r in zip(sample, res):
        g = gpu_lookup(q, s)
        if g is None:
            continue           # (a sampled cell the timed steps did not cover counts for nothing)
        covered_n += 1
        same += int((int(g["ref_aln_len"]), int(g["qry_aln_len"]), int(g["sim_errors"]), float(g["identity"]).hex(), int(g["status"])) ==
                    (int(r["ref_aln_len"]), int(r["qry_aln_len"]), int(r["sim_errors"]), float(r["identity"]).hex(), int(r["status"])))
    return {
        "value": (n_rel_job + n_unrel_job) / job_wall, "unit": "genome-pairs/s", "cores": threads, "kind": "port",
        "variant": "own-cpu (oracle/anim_cpu.cpp: the repo's CPU aligner — sparse 16-mer index of the reference (8 B per base, counting sort), every "
                   "5th query position looked up, mgaps clustering, postnuc extension with MUMmer's dynamic band and certified bands for the forced "
                   "re-alignments — g++ -O3 -mavx2, one pair per thread; NOT MUMmer's binary, whose suffix-tree matcher and full-rectangle "
                   "re-alignments cost more per pair)",
        "cpu_s_per_related_pair": t_rel, "cpu_s_per_unrelated_pair": t_unrel,
        "sample": note + f"{len(sample)} ordered pairs of the same job ({len(rel)} related, {len(unrel)} unrelated) run one per host "
                  f"thread ({min(threads, len(sample))} concurrent, {wall:.1f} s wall, {float(secs.sum()):.0f} CPU-s): "
                  f"{t_rel:.2f} s per related pair, {t_unrel:.2f} s per unrelated pair; extrapolated linearly to "
                  f"{n_rel_job} related + {n_unrel_job} unrelated pairs on {threads} threads",
        "job_seconds_extrapolated": job_wall, "cpu_seconds_extrapolated": job_cpu_s,
        "seconds_definition": "CPU seconds of the pair's own threads (CLOCK_THREAD_CPUTIME_ID: the pair's thread + the second strand's walker) — what a "
                              "pool of PROCESSES, pyani's runner, pays per pair.  Rounds 3-4 charged wall seconds inside a thread of one 128-thread "
                              "process, which includes the time a thread sleeps on the process-wide address-space lock while 127 others fault in "
                              "their 40 MB indexes (r04: 5.4 s per unrelated pair against 0.6 CPU-s) and made the baseline ~8 x slower than the box is",
        "gpu_parity_on_sample": f"{same}/{covered_n} sampled pairs (of {len(sample)}; the others were not among the timed cells) identical to the GPU's result",
    }


def _nucmer_job(job):
    ref, qry, prefix = job
    t0 = time.perf_counter()
    subprocess.run(["nucmer", "--mum", "-p", prefix, ref, qry], check=True, capture_output=True)
    with open(prefix + ".filter", "wb") as fh:
        subprocess.run(["delta-filter", "-1", prefix + ".delta"], check=True, stdout=fh, stderr=subprocess.DEVNULL)
    return time.perf_counter() - t0


def _nucmer_baseline(sample, n_rel, data, n_rel_job, n_unrel_job, threads):
    """pyani's own jobs (anim.py:240-289) under a multiprocessing.Pool(os.cpu_count()) (run_multiprocessing.py:130)."""
    import multiprocessing
    from pyani_amd import synth
    with tempfile.TemporaryDirectory() as tmp:
        paths = {}
        for g in sorted({g for p in sample for g in p}):
            paths[g] = os.path.join(tmp, f"{synth.genome_name(g)}.fna")
            synth.write_fasta(Path(paths[g]), data[g][0], data[g][1], synth.genome_name(g))
        jobs = [(paths[q], paths[s], os.path.join(tmp, f"{q}_vs_{s}")) for q, s in sample]
        t0 = time.perf_counter()
        with multiprocessing.Pool(threads) as pool:
            secs = pool.map(_nucmer_job, jobs)
        wall = time.perf_counter() - t0
    t_rel, t_unrel = float(np.mean(secs[:n_rel])), float(np.mean(secs[n_rel:]))
    job_cpu_s = n_rel_job * t_rel + n_unrel_job * t_unrel
    return {
        "value": (n_rel_job + n_unrel_job) / (job_cpu_s / threads), "unit": "genome-pairs/s", "cores": threads, "kind": "reference",
        "sample": f"{len(sample)} ordered pairs ({n_rel} related) through nucmer --mum + delta-filter -1 under Pool({threads}) "
                  f"({wall:.1f} s wall): {t_rel:.2f} s per related, {t_unrel:.2f} s per unrelated pair; extrapolated linearly",
        "job_seconds_extrapolated": job_cpu_s / threads, "cpu_seconds_extrapolated": job_cpu_s,
    }


def tetra_subrecord(eng, local, no_cpu):
    """C2 TETRA on the same GPU (200 genomes x 5 Mb, seed 20250228): the count kernel's HBM roofline, nested into the ANIm line."""
    from pyani_amd import _lib
    n, L, seed = 200, 5_000_000, 20250228
    data = synth_genomes(seed, n, L, 0, n, 1)
    ids = np.ascontiguousarray([eng.add_genome(s_, o_) for s_, o_ in data], dtype=np.int32)
    eng.upload()
    alg_bytes, bases = eng.tetra_algorithmic_bytes(ids.tolist())
    for _ in range(5):
        eng.tetra_matrix_enqueue(ids, fetch_z=False)
    eng.sync()
    eng.profile_reset()
    eng.profile_config(kernel_mask=1 << _lib.K_TETRA_COUNT, every_n=4)
    eng.profile_enable(True)
    steps = 40
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.tetra_matrix_enqueue(ids, fetch_z=False)
    eng.sync()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    ms, cnt = eng.profile_get(_lib.K_TETRA_COUNT)
    avg_s = ms / max(cnt, 1) * 1e-3
    pairs = n * (n - 1) // 2
    traffic = None
    pmc = ROOT / "profiles" / "pmc_tetra_count.json"
    if pmc.exists():
        traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
    achieved = alg_bytes / avg_s / 1e9 if cnt else 0.0
    rec = {
        "workload": f"C2: TETRA on {n} synthetic ~5 Mb genomes (seed {seed}), {pairs} unordered pairs, counts + Z + Pearson",
        "value": pairs / (elapsed / steps), "unit": "genome-pairs/s", "steps": steps, "ms_per_step": elapsed / steps * 1e3,
        "dtype": "u64 counts + f64 Z/Pearson",
        "roofline": {"bound": "hbm", "kernel": "tetra_count_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": int(alg_bytes),
                     "avg_launch_us": avg_s * 1e6, "launches": int(cnt), "frac_of_measured_copy_peak_6290": achieved / 6290.0},
    }
    if not no_cpu:
        eng.tetra_matrix_enqueue(ids, fetch_z=True)
        z, _, _ = eng.tetra_matrix_fetch(n)
        rec["cpu_baseline"] = tetra_cpu_baseline(z, data[:2], n, pairs)   # ids ascend in data order
    return rec


ASSEMBLY_S_PER_PAIR = 1.2e-6     # pyani_amd.anim.assemble_run_matrices, measured on the full C4 grid (999 000 pairs in 1.2 s)


def related_only_record(eng, args, stages=()):
    """One family of the same generator (25 genomes of one ancestor, all 600 ordered pairs related) in ONE call: the rate a
    genus-level job sees, where no pair is a cheap miss.  `steady`: four families (2400 related pairs) in one call — the same
    work with the launch tails of one call (the last forced run alone on the chip) amortised, as a 24 000-related-pair C4 grid
    sees them.  `stage_ms`: the 600-pair call once more with ONE worker and the stages' HIP events on (untimed)."""
    n, K = args.genomes, (args.genomes + 24) // 25
    fams = [[g for g in range(n) if g % K == f][:25] for f in (1, 2, 3, 4) if f < K]
    ids = []
    for fam in fams:
        data = [synth_genomes(args.seed, n, args.length, g, g + 1, 1)[0] for g in fam]
        ids.append([eng.add_genome(s_, o_) for s_, o_ in data])
    eng.upload()
    pairs = [(a, b) for a in ids[0] for b in ids[0] if a != b]
    many = [(a, b) for fam in ids for a in fam for b in fam if a != b]
    every = [(a, b) for fam in ids for a, b in zip(fam, fam[1:] + fam[:1])]
    eng.anim_pairs([a for a, _ in every], [b for _, b in every])      # seed lists built
    # each shape twice: the first call of a shape grows the workers' scratch to it (hipMalloc / hipFree of GBs: 0.2 - 0.7 s), which a
    # process pays once; the second is the rate it then sees.  Both are printed.
    t0 = time.perf_counter()
    eng.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = eng.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])
    dt = time.perf_counter() - t0
    rec = {"workload": f"{len(fams[0])} genomes of one ancestor ({len(pairs)} ordered pairs, all related), one call", "seconds": dt,
           "pairs_per_s": len(pairs) / dt, "pairs_with_alignment": int((res["status"] == 0).sum()), "first_call_seconds": first}
    if len(fams) > 1:
        t0 = time.perf_counter()
        eng.anim_pairs([a for a, _ in many], [b for _, b in many])
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.anim_pairs([a for a, _ in many], [b for _, b in many])
        dt = time.perf_counter() - t0
        rec["steady"] = {"workload": f"{len(fams)} such families in one call ({len(many)} ordered pairs, all related)", "seconds": dt,
                         "pairs_per_s": len(many) / dt, "first_call_seconds": first}
    if len(fams) > 1 and hasattr(eng, "anim_pairs_enqueue"):
        # the same four families as FOUR family-sized calls, two in flight at a time (pg_anim_pairs_enqueue / _fetch): what a caller that
        # submits one genus after another sees — the tail of each call overlaps the front of the next
        calls = [[(a, b) for a in fam for b in fam if a != b] for fam in ids]
        for rep in range(2):      # (first pass grows the lanes' scratch)
            t0 = time.perf_counter()
            pend = []
            for c in calls:
                pend.append(eng.anim_pairs_enqueue([a for a, _ in c], [b for _, b in c]))
                if len(pend) == 2:
                    eng.anim_pairs_fetch(pend.pop(0))
            for t in pend:
                eng.anim_pairs_fetch(t)
            dt = time.perf_counter() - t0
        rec["pipelined_calls"] = {"workload": f"{len(calls)} family-sized calls ({len(calls[0])} pairs each), two in flight (pg_anim_pairs_enqueue / _fetch)",
                                  "seconds": dt, "pairs_per_s": sum(len(c) for c in calls) / dt, "seconds_per_call": dt / len(calls)}
    if stages:
        eng.anim_set_workers(1)
        eng.profile_reset()
        eng.profile_config(kernel_mask=sum(1 << s_ for s_ in stages), every_n=1)
        eng.profile_enable(True)
        t0 = time.perf_counter()
        eng.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])
        eng.sync()
        rec["one_worker_seconds"] = time.perf_counter() - t0
        eng.profile_enable(False)
        rec["stage_ms"] = {eng.kernel_name(s_): round(eng.profile_get(s_)[0], 3) for s_ in stages}
        eng.anim_set_workers(2)
    return rec


def unrelated_only_record(eng, args, ids):
    """The other half of the job on its own: one genome of each of (up to) 100 different ancestors, every ordered pair between
    them — no pair has an alignment — in ONE call (genomes already resident, seed lists built by the timed steps)."""
    n, K = args.genomes, (args.genomes + 24) // 25
    pick = [ids[g] for g in range(min(K, 100))]            # genomes 0 .. K-1 descend from K different ancestors
    if len(pick) < 3:
        return None
    pairs = [(a, b) for a in pick for b in pick if a != b]
    eng.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])                # (lists of these genomes built)
    t0 = time.perf_counter()
    res = eng.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])
    dt = time.perf_counter() - t0
    return {"workload": f"{len(pick)} genomes of {len(pick)} different ancestors ({len(pairs)} ordered pairs, none related), one call", "seconds": dt,
            "pairs_per_s": len(pairs) / dt, "pairs_with_alignment": int((res["status"] == 0).sum())}


SIMDS, CLOCK_GHZ = 1024, 2.4          # MI355X: 256 CUs x 4 SIMDs; a SIMD issues one wave64 VALU instruction per 4 cycles


def _side_record(fn, *a):
    """A side record (outside the timed region) must never cost the headline line: its failure is reported in its place."""
    try:
        return fn(*a)
    except Exception as e:  # noqa: BLE001 - reported, not hidden
        return {"error": f"{type(e).__name__}: {e}"}


def sketch_record(eng, ids, n, K, dense, covered):
    """The opt-in SKETCH mode (fastANI-shaped estimate, pg_sketch_pairs) on the WHOLE grid of the same job, for scale — an
    estimate with its own columns, not what `value` measures: ordered pairs per second with the sketches resident, the one-off sketch
    build, and its agreement with the exact engine's identity over the related pairs this run computed."""
    ids = np.asarray(ids, dtype=np.int32)
    q = np.repeat(np.arange(n), n)
    r = np.tile(np.arange(n), n)
    keep = q != r
    q, r = q[keep], r[keep]
    t0 = time.perf_counter()
    eng.sketch_pairs(ids[:n], ids[:n])                  # builds every genome's sketch (cached), n pairs
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = eng.sketch_pairs(ids[q], ids[r])              # (query fragmented, reference as a k-mer set)
    dt = time.perf_counter() - t0
    related = (q % K) == (r % K)
    got = res["status"] == 0
    # exact identity of nucmer(reference = a, query = b) sits in dense[a, b]; the sketch's query is nucmer's query
    ex_ok = covered[r, q] & (dense[r, q, 5] == 0)
    both = related & got & ex_ok
    err = np.abs(res["ani"][both] - dense[r, q, 4][both].view(np.float64)) if both.any() else np.zeros(0)
    ex_id = dense[r, q, 4][both].view(np.float64) if both.any() else np.zeros(0)
    return {"pairs": int(len(q)), "seconds": dt, "pairs_per_s": len(q) / dt, "sketch_build_s": t_build, "frag_len": 3000, "scale": 16, "min_fraction": 0.2,
            "related_pairs_with_result": int((related & got).sum()), "unrelated_pairs_with_result": int((~related & got).sum()),
            "vs_exact_identity": None if not len(err) else {
                "pairs": int(len(err)), "mean_abs_error": float(err.mean()), "max_abs_error": float(err.max()),
                "max_abs_error_identity_ge_0.90": float(err[ex_id >= 0.9].max()) if (ex_id >= 0.9).any() else None},
            "note": "pg_sketch_pairs (FracMinHash containment per 3 kb query fragment, k = 16): an ESTIMATE in fastANI's output shape "
                    "(pyani/fastani.py:193-270), never mixed into the exact matrices; not what `value` measures"}


def _pmc_profile():
    f = ROOT / "profiles" / "pmc_anim.json"
    return json.loads(f.read_text()) if f.exists() else {}


def run_anim_cold(args, local):
    """ONE cold end-to-end run, one wall clock: FASTA files on disk -> pg_add_fasta_batch (read + parse +
    2-bit pack on the host threads) -> upload -> the whole N x N grid (seed lists built on first use) in steps of R rows ->
    pyani_amd.anim.assemble_run_matrices -> run_matrices_to_json.  Writing the synthetic FASTA files is NOT timed (they are
    the job's input)."""
    from pyani_amd import anim, parallel, synth
    from pyani_amd.engine import Engine
    n, R = args.genomes, max(1, min(args.rows_per_step, args.genomes))
    tmp = Path(tempfile.mkdtemp(prefix="pyani_cold_", dir=os.environ.get("PYANI_BENCH_TMP", None)))
    try:
        t_write = time.perf_counter()
        with ThreadPoolExecutor(min(64, os.cpu_count() or 2)) as ex:
            def put(g):
                seq, off = synth.genome(args.seed, n, g, args.length)
                f = tmp / f"{synth.genome_name(g)}.fna"
                synth.write_fasta(f, seq, off, synth.genome_name(g))
                return f
            paths = list(ex.map(put, range(n)))
        t_write = time.perf_counter() - t_write
        nbytes = sum(f.stat().st_size for f in paths)
        t0 = time.perf_counter()
        eng = Engine(local)
        added = eng.add_fasta_batch(paths)
        t_ingest = time.perf_counter() - t0
        eng.upload()
        eng.sync()
        t_upload = time.perf_counter() - t0 - t_ingest
        ids = np.asarray([a[0] for a in added], dtype=np.int32)
        stems = [f.stem for f in paths]
        lengths = {stems[k]: int(added[k][1]) for k in range(n)}
        res = {}
        t_grid = time.perf_counter()
        for k in range((n + R - 1) // R):
            rows = [r for r in range(k * R, min(n, (k + 1) * R))]
            pairs = parallel.anim_pair_array(n, rows, symmetric=True)
            out = eng.anim_pairs(ids[pairs[:, 0]], ids[pairs[:, 1]])
            ok = out["status"] == 0
            for (q, s_), r in zip(pairs[ok].tolist(), out[ok]):
                res[(stems[q], stems[s_])] = (int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]), int(r["sim_errors"]))
        t_grid = time.perf_counter() - t_grid
        t_asm = time.perf_counter()
        mats = anim.assemble_run_matrices(res, lengths)
        js = anim.run_matrices_to_json(mats)
        t_asm = time.perf_counter() - t_asm
        total = time.perf_counter() - t0
        eng.close()
        print(json.dumps({
            "metric": "end-to-end cold wall-clock for the N x N ANIm matrices, from FASTA files on disk to the run's five matrices as JSON",
            "value": total, "unit": "s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": total * 1e3, "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32 packed DP words, i64 lengths, f64 identity", "data": "synthetic",
            "end_to_end_cold_s": total,
            "config": {"workload": f"C4 cold: {n} synthetic ~{args.length / 1e6:g} Mb FASTA files ({nbytes / 1e9:.2f} GB of text) -> {n * (n - 1)} ordered pairs -> matrices",
                       "genomes": n, "fasta_bytes": nbytes, "pairs_with_alignment": len(res),
                       "seconds": {"read_parse_pack": t_ingest, "upload": t_upload, "grid": t_grid, "matrices_and_json": t_asm},
                       "json_bytes": sum(len(v) for v in js.values()), "not_timed_writing_the_input_files_s": t_write,
                       "host_threads": os.cpu_count()}}), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


REHEARSAL = os.environ.get("PYANI_BENCH_REHEARSAL") == "1"


class _RehearsalEngine:
    """PYANI_BENCH_REHEARSAL=1 only (tests/test_parallel_gloo.py): a stand-in for pyani_amd.engine.Engine that COMPUTES NOTHING — it
    sleeps what a call's pairs would cost (a related pair ~130 x an unrelated one, 4 x apart by family, as measured on C4) and returns records that are a function of the pair — so that the whole N > 1 control flow of this file (RowQueue dealing,
    anim_allgather_dynamic, the strong-step series, the imbalance figures, the result hash) can be run with 8 gloo ranks on a box
    without a GPU.  A rehearsal's line says so in `data` and carries no roofline; it is never what the driver measures."""

    def __init__(self, n):
        self.K = (n + 24) // 25

    def add_genome(self, seq, off):
        self._n = getattr(self, "_n", 0) + 1
        return self._n - 1

    def upload(self): pass
    def sync(self): pass
    def close(self): pass
    def anim_set_workers(self, w): pass
    def anim_set_batch_budget(self, a, b): pass
    def profile_reset(self): pass
    def profile_config(self, **kw): pass
    def profile_enable(self, on=True): pass
    def profile_get(self, which): return (0.0, 0)
    def kernel_name(self, which): return f"stage{which}"
    def anim_counters(self, reset=False): return np.zeros(64, dtype=np.uint64)

    def anim_pairs(self, ref_ids, qry_ids, **kw):
        from pyani_amd.engine import Engine
        r, q = np.asarray(ref_ids, dtype=np.int64), np.asarray(qry_ids, dtype=np.int64)
        rel = (r % self.K) == (q % self.K)
        # C4 on MI355X: ~4.6 us per unrelated ordered pair, ~0.6 ms per related one on average, 0.3 - 1.2 ms by the divergence of the
        # family members (a row = ~975 unrelated + 0 ... 48 related ordered pairs: 5 - 60 ms, a 100-row share ~1.8 s); here at 1/5 scale
        fam_cost = 0.5 + 1.5 * ((np.minimum(r, q)[rel] // self.K) % 6) / 5.0
        time.sleep(float(os.environ.get("PYANI_BENCH_REHEARSAL_SCALE", "1")) * (1.2e-4 * float(fam_cost.sum()) + 1e-6 * float((~rel).sum())))
        out = np.zeros(len(r), dtype=Engine.ANIM_DTYPE)
        out["ref_aln_len"], out["qry_aln_len"], out["sim_errors"] = r * 1000 + q, q * 1000 + r, (r * 7 + q * 13) % 1000
        out["n_alignments"] = np.where(rel, 3, 0)
        out["identity"] = np.where(rel, 0.9 + 1e-4 * ((r + q) % 100), 0.0)
        out["status"] = np.where(rel, 0, 1)
        return out


def run_anim(args, rank, world, local, dist, torch):
    from pyani_amd import _lib, parallel
    from pyani_amd.engine import Engine
    if args.cold_e2e:
        return run_anim_cold(args, local)
    eng = _RehearsalEngine(args.genomes) if REHEARSAL else Engine(local)
    n, R = args.genomes, max(1, min(args.rows_per_step, args.genomes))
    if os.environ.get("PYANI_BENCH_BATCH_PAIRS"):   # development: pairs / matches per internal launch of pg_anim_pairs
        eng.anim_set_batch_budget(int(os.environ["PYANI_BENCH_BATCH_PAIRS"]), int(os.environ.get("PYANI_BENCH_BATCH_MATCHES", 256 << 20)))
    K = (n + 24) // 25                       # ancestors of the SURVEY.md §8(d) generator: genome g descends from ancestor g mod K
    t_prep = time.perf_counter()
    data = synth_genomes(args.seed, n, args.length, 0, n, world)
    ids = [eng.add_genome(s_, o_) for s_, o_ in data]
    eng.upload()
    t_prep = time.perf_counter() - t_prep
    dev = torch.device("cpu") if REHEARSAL else torch.device("cuda", local)
    lens = np.array([len(d[0]) for d in data], dtype=np.int64)
    # N > 1: a step's rows are dealt over the ranks in a fixed scrambled order, ONE engine call per rank and step (measured on
    # MI355X, profiles/r05_deal_probe.json: the 8 shares of a C4 step are within 2 - 5 % of each other, while a call of 6 rows costs
    # 2.5 x as much per row as a call of 100 — every call pays the tails of its kernels — so round 4's dynamic deal ran at 48 % of
    # the plain rate).  --dynamic-deal: the ranks PULL guided chunks from a cross-rank counter (pyani_amd.parallel.RowQueue), for
    # jobs whose cost sits in a few rows.
    queue = None
    if dist is not None and args.dynamic_deal:
        queue = parallel.RowQueue(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"))      # (the counter sits in the job's own rendezvous store)

    def rows_of(step, rows_per_step=None):
        r = rows_per_step or R
        return [(step * r + i) % n for i in range(r)]

    ids_np = np.asarray(ids, dtype=np.int32)

    # --pipeline (round 6; measured and NOT made the default, see the flag's help): before a step's call is fetched the next step's call is already enqueued
    # (pg_anim_pairs_enqueue / _fetch: two calls in flight on disjoint worker slots), so the sequential tail of one step — one forced
    # re-alignment can hold a single wave for 0.1 s — overlaps the front of the next, as the pool of pyani's runner keeps its cores busy
    # across job boundaries (run_multiprocessing.py:130-144).  Every step is still computed in full inside the timed region (the closing
    # fence waits for the last one).  Default: one blocking call per step.
    pipeline = (not REHEARSAL) and args.pipeline and queue is None
    prefetched = {}
    upcoming = [None]      # the pairs this rank will be asked for next (set by step())

    def _key(pairs):
        return (len(pairs), int(pairs[0, 0]), int(pairs[0, 1]), int(pairs[-1, 0]), int(pairs[-1, 1])) if len(pairs) else None

    def compute(pairs):   # pairs: int64 [m, 2] of (reference, query) genome numbers
        if not pipeline or not len(pairs):
            return parallel.anim_records_to_tensor(eng.anim_pairs(ids_np[pairs[:, 0]], ids_np[pairs[:, 1]]), dev)
        t = prefetched.pop(_key(pairs), None)
        if t is None:
            t = eng.anim_pairs_enqueue(ids_np[pairs[:, 0]], ids_np[pairs[:, 1]])
        nxt, upcoming[0] = upcoming[0], None
        if nxt is not None and len(nxt) and _key(nxt) not in prefetched:
            prefetched[_key(nxt)] = eng.anim_pairs_enqueue(ids_np[nxt[:, 0]], ids_np[nxt[:, 1]])
        return parallel.anim_records_to_tensor(eng.anim_pairs_fetch(t), dev)

    tiles, imbalance = {}, []

    def step(k, keep=False, rows=None, key=None, then=None):
        rows = rows_of(k) if rows is None else rows
        if pipeline and then is not None:      # this rank's share of the NEXT step of the same loop
            upcoming[0] = parallel.anim_pair_array(n, rows_of(then) if dist is None else parallel.anim_row_shard(rows_of(then), rank, world), symmetric=True)
        pairs = parallel.anim_pair_array(n, rows, symmetric=True)   # the rows' unordered pairs, both directions
        if dist is None:
            vals = compute(pairs)
        else:
            if queue is not None:
                grid, st = parallel.anim_allgather_dynamic(compute, n, dev, queue, key or f"k{k}", rows, symmetric=True)
                if keep:
                    imbalance.append(st)
            else:
                st = {}
                grid = parallel.anim_allgather(compute, n, dev, rows=rows, symmetric=True, stats=st)
                if keep:
                    imbalance.append(st)
            vals = grid[torch.from_numpy(pairs[:, 0]).to(dev), torch.from_numpy(pairs[:, 1]).to(dev)]
        if keep:
            tiles[k] = (pairs, vals)
        return pairs

    def fence():
        eng.sync()
        if not REHEARSAL:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not REHEARSAL:
            torch.cuda.synchronize()

    t_cold = time.perf_counter()
    for k in range(args.warmup):
        step(k)
        if k == 0:
            fence()
            t_cold = time.perf_counter() - t_cold      # the first step also builds the seed lists of every genome it touches
    fence()
    if args.warmup == 0:
        t_cold = None
    stages = [_lib.K_ANIM_SEED, _lib.K_ANIM_HIT, _lib.K_ANIM_CLUSTER, _lib.K_ANIM_GAPS, _lib.K_ANIM_FWD, _lib.K_ANIM_BWD, _lib.K_ANIM_EXTEND,
              _lib.K_ANIM_EXTLANE, _lib.K_ANIM_FINISH]
    ext_stages = [_lib.K_ANIM_GAPS, _lib.K_ANIM_FWD, _lib.K_ANIM_BWD, _lib.K_ANIM_EXTEND, _lib.K_ANIM_EXTLANE]
    # ---- the timed region: K steps between fences, no per-kernel events (they belong to the one-worker step below) ----------
    t0 = time.perf_counter()
    last = args.warmup + args.steps - 1
    for k in range(args.warmup, args.warmup + args.steps):
        step(k, keep=True, then=k + 1 if k < last else None)
    fence()
    elapsed = time.perf_counter() - t0
    assert not prefetched, "a prefetched step was never fetched"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the roofline pass (untimed): ONE worker, so that the stages' HIP events are not timing the other worker's persistent waves
    # (two streams made `anim_finish_kernel` look 379 ms long), over EVERY tile of the grid (one tile is
    # not representative — the ten tiles of C4 hold 1.1 ... 2.5 x 10^12 DP cells), engine counters read around each.  At N > 1 each
    # rank runs its static share of two tiles; rank 0 reports its own.
    k_prof = args.warmup + args.steps
    n_tiles = max(1, n // R)
    prof_tiles = list(range(n_tiles)) if world == 1 else [k_prof % n_tiles, (k_prof + 1) % n_tiles]
    if args.roofline_tiles:
        prof_tiles = prof_tiles[:args.roofline_tiles]
    eng.anim_set_workers(1)
    eng.profile_config(kernel_mask=sum(1 << s_ for s_ in stages), every_n=1)
    tile_recs = []
    for tno in prof_tiles:
        pp = parallel.anim_pair_array(n, parallel.anim_row_shard(rows_of(tno), rank, world), symmetric=True)
        eng.anim_counters(reset=True)
        eng.profile_reset()
        eng.profile_enable(True)
        t1 = time.perf_counter()
        compute(pp)
        eng.sync()
        dt = time.perf_counter() - t1
        eng.profile_enable(False)
        tile_recs.append({"tile": tno, "pairs": pp, "seconds": dt, "prof": {eng.kernel_name(s_): eng.profile_get(s_) for s_ in stages},
                          "ext_ms": sum(eng.profile_get(s_)[0] for s_ in ext_stages), "cnt": eng.anim_counters()})
    prof_pairs = np.concatenate([t_["pairs"] for t_ in tile_recs])
    prof_s = sum(t_["seconds"] for t_ in tile_recs)
    prof = {name: (sum(t_["prof"][name][0] for t_ in tile_recs), sum(t_["prof"][name][1] for t_ in tile_recs)) for name in tile_recs[0]["prof"]}
    ext_ms = sum(t_["ext_ms"] for t_ in tile_recs)
    cnt = np.sum([t_["cnt"] for t_ in tile_recs], axis=0)
    eng.anim_set_workers(2)
    fence()

    # ---- N > 1: the same tile size as N = 1's step, dealt over all ranks (strong scaling of ONE step: launches shrink with N)
    strong = None
    bare = args.no_side_records or (args.no_cpu_baseline and not args.side_records)      # (tests and profiling runs: the step loop and the roofline step only)
    if dist is not None and (not bare or REHEARSAL):
        r1 = max(1, n // 10)
        step(k_prof + 1, rows=rows_of(k_prof + 1, r1), key="strong_warm")
        fence()
        ts = time.perf_counter()
        np_strong = 0
        for j in range(2):
            np_strong += len(step(k_prof + 2 + j, rows=rows_of(k_prof + 2 + j, r1), key=f"strong{j}"))
        fence()
        ts = time.perf_counter() - ts
        tt = torch.tensor([ts], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        strong = {"rows_per_step": r1, "steps": 2, "ms_per_step": float(tt.item()) / 2 * 1e3, "pairs_per_s": np_strong / float(tt.item()),
                  "note": f"the N = 1 step ({r1} rows, ~{r1 * (n - 1)} ordered pairs) dealt over {world} ranks: each rank's launches are 1/{world} the size"}

    if rank == 0:
        P = np.concatenate([tiles[k][0] for k in sorted(tiles)])                      # [M, 2] (reference, query)
        g = torch.cat([tiles[k][1] for k in sorted(tiles)]).cpu().numpy()           # [M, FIELDS]
        pairs_done = len(P)
        related_m = (P[:, 0] % K) == (P[:, 1] % K)
        status = g[:, 5]
        ident = g[:, 4].view(np.float64)
        n_related = int(related_m.sum())
        ok_rel = int(((status == 0) & related_m).sum())
        unrel_aln = int(((status == 0) & ~related_m).sum())
        step_s = elapsed / args.steps
        # byte roofline of the dominant stage of the one-worker step
        alg_prof = float(((lens[prof_pairs[:, 0]] + 3) // 4 + (lens[prof_pairs[:, 1]] + 3) // 4 + 32).sum())
        alg_bytes = float(((lens[P[:, 0]] + 3) // 4 + (lens[P[:, 1]] + 3) // 4 + 32).sum())
        dom = max(prof, key=lambda name: prof[name][0])
        dom_ms, dom_n = prof[dom]
        achieved = alg_prof / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
        pmc = _pmc_profile()
        traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch")
        # per tile: the dominant stage's time and byte rate (the tiles differ ~2 x in DP cells: the extremes are printed, `achieved` is the
        # mean over all of them = total algorithmic bytes / total time of that stage)
        def _alg(pp):
            return float(((lens[pp[:, 0]] + 3) // 4 + (lens[pp[:, 1]] + 3) // 4 + 32).sum())
        per_tile = [{"tile": t_["tile"], "pairs": int(len(t_["pairs"])), "ms": round(t_["prof"][dom][0], 3),
                     "GBps": round(_alg(t_["pairs"]) / (t_["prof"][dom][0] * 1e-3) / 1e9, 2) if t_["prof"][dom][0] else None,
                     "dp_cells": int(t_["cnt"][2]) + int(t_["cnt"][5]) + int(t_["cnt"][8]), "kernel_ms_sum": round(sum(v[0] for v in t_["prof"].values()), 1),
                     "stage_ms": {name_: round(v[0], 1) for name_, v in t_["prof"].items()}}
                    for t_ in tile_recs]
        # VALU-issue roofline of the extension stage (integer DP in registers: bytes are not its bound): DP cells per second against the
        # rate at which the chip can issue the vector instructions those cells cost.  MEASURED IN THIS RUN: cells, anti-diagonals and calls
        # per kernel class (the engines' own counters, pg_anim_counters) and the stages' HIP-event times of the roofline pass above.
        # FROM THE COMMITTED PROFILE (profiles/pmc_anim.json: a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU pass of the same command on
        # MI355X, tools/summarize_round_profiles.py): instructions per cell, per kernel class and for the stage as a whole.
        cells = int(cnt[2]) + int(cnt[5]) + int(cnt[8])
        vpc, spc = pmc.get("extension_valu_per_cell"), pmc.get("extension_salu_per_cell")
        kclass = [("gaps", "anim_postnuc_gap_kernels", 128), ("forward", "anim_postnuc_fwd_kernel", 128), ("backward_ahead", "anim_postnuc_rehearse_kernel+anim_postnuc_bwd_kernel", 128),
                  ("walks", "anim_postnuc_kernel", 128), ("forced_narrow", "anim_postnuc_forced_kernels", None), ("forced_512_1024", "anim_postnuc_forced_kernels", None),
                  ("forced_2048", "anim_postnuc_forced_kernels", None), ("forced_group", "anim_postnuc_forced_kernels", None)]
        per_kernel = {}
        for kc, (kname, stage_name, slots) in enumerate(kclass):
            calls, steps_k, cells_k = (int(cnt[32 + 4 * kc + j]) for j in range(3))
            if not calls:
                continue
            rec = {"calls": calls, "anti_diagonals": steps_k, "cells": cells_k, "stage": stage_name, "stage_ms": round(prof.get(stage_name, (0.0, 0))[0], 3)}
            if slots:      # a trimmed search runs on the 256-diagonal window: 64 lanes x 2 cell slots per anti-diagonal
                rec["live_slot_fraction"] = round(cells_k / (steps_k * float(slots)), 4) if steps_k else None
            ipc_k = (pmc.get("valu_per_cell_by_class") or {}).get(kname)
            if ipc_k:
                rec["valu_instructions_per_cell"] = ipc_k
            per_kernel[kname] = rec
        # instructions per CU and cycle of a stage = its cells (this run) x instructions per cell (committed profile) / its event time (this run)
        for stage_name in {r_["stage"] for r_ in per_kernel.values()}:
            ks = [r_ for r_ in per_kernel.values() if r_["stage"] == stage_name and r_.get("valu_instructions_per_cell")]
            ms_ = prof.get(stage_name, (0.0, 0))[0]
            if ks and ms_ and len(ks) == sum(1 for r_ in per_kernel.values() if r_["stage"] == stage_name):
                instr = sum(r_["cells"] * r_["valu_instructions_per_cell"] for r_ in ks)
                for r_ in ks:
                    r_["stage_valu_per_cu_cycle"] = round(instr / 256.0 / (ms_ * 1e-3 * CLOCK_GHZ * 1e9), 3)
        valu = None
        if vpc and ext_ms:
            # a CU issues one wave64 VALU instruction per cycle (4 SIMDs x 1 per 4 cycles) and one scalar instruction per cycle (ONE scalar
            # unit per CU): the step loop of the engines spends about as many scalar as vector instructions, so whichever is larger binds
            ipc = max(vpc, spc or 0.0)
            peak_cells = SIMDS * CLOCK_GHZ * 1e9 / 4.0 / ipc
            valu = {"bound": "valu-issue" if ipc == vpc else "salu-issue",
                    "kernels": "anim_postnuc_{gaplane,gapbig,fwd,rehearse,bwd,(walk),forced,forced_wide,forced_huge}_kernel",
                    "achieved": cells / (ext_ms * 1e-3), "peak": peak_cells, "unit": "DP cells/s", "frac": cells / (ext_ms * 1e-3) / peak_cells,
                    "cells": cells, "anti_diagonals": int(cnt[1]), "extension_ms": ext_ms, "valu_instructions_per_cell": vpc,
                    "salu_instructions_per_cell": spc, "instructions_per_cell_replayed": True,
                    "measured_in_this_run": "cells, anti_diagonals, extension_ms, per_kernel.{calls, anti_diagonals, cells, stage_ms, live_slot_fraction} (engine counters + HIP events of the one-worker pass over all tiles)",
                    "from_committed_profile": f"valu / salu instructions per cell, per_kernel.valu_instructions_per_cell ({pmc.get('extension_valu_source', 'profiles/pmc_anim.json')})",
                    "per_kernel": per_kernel,
                    "peak_definition": f"256 CUs x {CLOCK_GHZ} GHz x 1 wave64 VALU instruction per cycle ({SIMDS} SIMDs, 4 cycles each) or 1 scalar "
                                       f"instruction per cycle (one scalar unit per CU) / instructions per cell of the binding kind"}
        # hash of one whole N x N result grid (the last occurrence of every cell among the timed steps), if the steps cover it
        dense = np.zeros((n, n, g.shape[1]), dtype=np.int64)
        covered = np.zeros((n, n), dtype=bool)
        dense[P[:, 0], P[:, 1]] = g
        covered[P[:, 0], P[:, 1]] = True
        sha = hashlib.sha1(dense.tobytes()).hexdigest() if int(covered.sum()) == n * (n - 1) else None
        grid_s = elapsed / pairs_done * n * (n - 1)
        # ---- the line checks itself: (1) the cells of the grid just computed that are ALSO pairs of the committed goldens of
        # the INDEPENDENT oracle (tests/golden/anim_oracle_family_digests.json.gz: two whole C4 families = 1 200 ordered pairs, made by
        # oracle/nucmer_oracle.cpp + oracle/anim_oracle.py, which share no header with the engine) must hold the oracle's tuple bit for bit;
        # (2) the whole-grid hash must be the committed one.  Both only read committed DATA (no oracle code runs here).
        parity_indep, sha_check = None, None
        if not REHEARSAL:
            try:
                import gzip
                with gzip.open(ROOT / "tests" / "golden" / "anim_oracle_family_digests.json.gz", "rt") as fh:
                    fam = json.load(fh)
                if (fam["n"], fam["L"], fam["seed"]) == (n, args.length, args.seed):
                    same = checked = 0
                    first_bad = None
                    for a_, b_, _nrec, _nkept, _h1, _h2, tup in fam["pairs"]:
                        if not covered[a_, b_]:
                            continue
                        checked += 1
                        c = dense[a_, b_]
                        got_t = None if int(c[3]) == 0 else [int(c[0]), int(c[1]), float(np.int64(c[4]).view(np.float64)).hex(), int(c[2]), int(c[3])]
                        if got_t == tup:
                            same += 1
                        elif first_bad is None:
                            first_bad = {"ref": a_, "qry": b_, "gpu": got_t, "oracle": tup}
                    parity_indep = {"identical": same, "checked": checked, "golden_pairs": len(fam["pairs"]), "first_difference": first_bad,
                                    "what": "(ref_aln_len, qry_aln_len, identity bits, sim_errors, n_alignments) of pg_anim_pairs, filter on, against the committed goldens of the "
                                            "independent oracle (oracle/nucmer_oracle.cpp + oracle/anim_oracle.py; tools/make_anim_family_hashes.py)"}
            except Exception as exc:  # noqa: BLE001
                parity_indep = {"error": repr(exc)}
            want_sha = args.expect_sha
            if want_sha == "auto":
                f_ = ROOT / "tests" / "golden" / "anim_c4_grid_sha1.txt"
                default_job = (n, args.length, args.seed) == (1000, 5_000_000, 20250301)
                want_sha = f_.read_text().split()[0] if (default_job and f_.exists()) else "none"
            if want_sha != "none" and sha is not None:
                sha_check = {"expected": want_sha, "ok": sha == want_sha}
        measured_cold = None       # (never replayed: `bench.py --cold-e2e` measures it; profiles/ holds the last measurement)
        out = {
            "metric": "genome-pairs/sec (ordered pairs) + wall-clock for the N x N ANIm grid: nucmer --mum + delta-filter -1 + "
                      "parse_delta equivalent per ordered pair, genomes resident in HBM; vs the CPU path on this box's host cores",
            "value": pairs_done / elapsed, "unit": "genome-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak" if args.rows_default else "strong", "vs_baseline": None,
            "dtype": "u32 packed DP words (score << 17 | state << 15 | errors), i64 lengths, f64 identity",
            "data": "REHEARSAL: no engine, no GPU — a stub that sleeps a cost model (PYANI_BENCH_REHEARSAL=1; control-flow test only)" if REHEARSAL else "synthetic",
            "related_pairs_per_s": n_related / elapsed,
            "series": {
                "weak": {"rows_per_step_per_gpu": R // world if args.rows_default else None, "pairs_per_s": pairs_done / elapsed, "ms_per_step": step_s * 1e3},
                "fixed_grid": {"grid_pairs": n * (n - 1), "wall_s_grid": grid_s,
                               "note": "the FIXED 1000-genome grid at this rate: strong-scaling speed-up over N = 1 = wall_s_grid(1) / wall_s_grid(N) "
                                       "(per-GPU work per step is constant, so the weak series prices the fixed grid directly)"},
                "strong_step": strong,
            },
            "imbalance": None if not imbalance else {
                "max_over_mean_rank_busy_time_per_step": [round(st["imbalance"], 4) for st in imbalance],
                "mean": float(np.mean([st["imbalance"] for st in imbalance])), "worst": float(max(st["imbalance"] for st in imbalance)),
                "chunks_per_rank_last_step": imbalance[-1]["chunks"], "host_ms_last_step": imbalance[-1].get("host_ms"), "dealing": (f"guided chunks from a cross-rank counter (pyani_amd.parallel.RowQueue, {queue.kind})" if queue else
                                                                        "fixed scrambled deal of the step's rows, one engine call per rank and step (pyani_amd.parallel.anim_row_shard)")},
            "config": {
                "workload": f"C4: ANIm N x N grid on {n} synthetic ~{args.length / 1e6:g} Mb genomes (SURVEY.md §8(d) generator, seed "
                            f"{args.seed}; {n * (n - 1)} ordered pairs, {n * (n // K - 1)} of them between descendants of one ancestor); "
                            + (f"a step = the whole grid" if R == n else
                               f"a step = the unordered pairs owned by {R} genomes, in both directions (pyani_amd.parallel."
                               f"anim_pair_array(symmetric=True): ~{R * (n - 1)} ordered pairs), "
                               f"{n // R if n % R == 0 else n / R:g} steps = the whole grid"),
                "genomes": n, "rows_per_step": R, "pairs_per_step": pairs_done / args.steps, "pairs_timed": pairs_done,
                "related_pairs_timed": n_related, "related_pairs_with_alignment": ok_rel, "unrelated_pairs_with_alignment": unrel_aln,
                "grid_pairs": n * (n - 1), "wall_s_grid": grid_s,
                "identity_related_min_med_max": [float(x) for x in np.percentile(ident[(status == 0) & related_m], [0, 50, 100])]
                if ok_rel else None,
                "step_loop": ("pipelined: step k + 1 enqueued before step k is fetched (pg_anim_pairs_enqueue / _fetch, two calls in flight)" if pipeline
                              else "one blocking pg_anim_pairs per step"),
                "results_sha1_full_grid": sha, "results_sha1_check": sha_check, "parity_vs_independent_oracle": parity_indep,
                "parallelism": (f"1 process/GPU x {world}; genomes replicated; each step's rows "
                                + ("pulled in guided chunks from a cross-rank counter" if queue is not None else "dealt over the ranks by a fixed hash")
                                + "; one RCCL all-gather of 64 B per pair per step") if world > 1 else "1 GPU",
                "host_prep_s": t_prep, "cold_first_step_s": t_cold,
                "end_to_end_cold_s_estimate": (t_prep + (t_cold or step_s) + (n / R - 1) * step_s + ASSEMBLY_S_PER_PAIR * n * (n - 1)),
                "end_to_end_cold_s_measured": measured_cold,
                "note_end_to_end": "estimate = this run's synthetic-genome generation + packing + upload + first step + the remaining steps of one grid at "
                                   "the timed rate + matrix assembly; measured = `bench.py --cold-e2e` (FASTA files on disk -> matrices as JSON, one wall "
                                   "clock), kept in profiles/pmc_anim.json with its provenance",
                "extender": "nucmer (MUMmer 3.23's postnuc algorithm restated: exact on every nucmer output file the reference's tests hold)",
                "note_related_pairs_per_s": "related pairs timed / the same wall time (unrelated pairs of the tile included): a lower "
                                            "bound of the related-only rate; 97.6 % of C4's pairs are unrelated by construction",
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_replayed": True,
                "traffic_source": "profiles/pmc_anim.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (counters cannot be read inside a timed run); replayed, not measured in this run",
                "algorithmic_bytes_per_launch": alg_prof / max(dom_n, 1), "avg_launch_ms": dom_ms / max(dom_n, 1),
                "launches": int(dom_n),
                "definition": "SURVEY.md §8(d): ceil(Lq/4) + ceil(Ls/4) + 32 B per ordered pair, summed over an untimed pass of EVERY tile of the grid "
                              "run with a single worker (stage events un-overlapped; rank 0's share of two tiles at N > 1) / the summed HIP-event "
                              "time of the stage that took longest in it; per_tile has each tile, tile_min / tile_max the extremes",
                "tiles": len(tile_recs), "per_tile": per_tile,
                "tile_min_GBps": min((t_["GBps"] for t_ in per_tile if t_["GBps"]), default=None),
                "tile_max_GBps": max((t_["GBps"] for t_ in per_tile if t_["GBps"]), default=None),
                "one_worker_step": {"pairs": int(len(prof_pairs)), "seconds": prof_s, "tiles": len(tile_recs),
                                    "stage_ms": {name: round(ms, 3) for name, (ms, _) in prof.items()},
                                    "stage_launches": {name: int(c) for name, (_, c) in prof.items()},
                                    "note": "seconds = wall time of the call, which includes growing the single worker's scratch to the whole "
                                            "launch budget (it held half of it while two workers ran) and the event synchronisations; the "
                                            "kernels' own time is the sum of stage_ms"},
                "pipeline_achieved": alg_bytes / elapsed / 1e9, "pipeline_frac": alg_bytes / elapsed / 1e9 / HBM_PEAK_GBS,
                "timed_region_ms": round(elapsed * 1e3, 3),
                "valu_issue": valu,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_lookup(q, s_):
                if not covered[q, s_]:
                    return None
                c = dense[q, s_]
                return {"ref_aln_len": c[0], "qry_aln_len": c[1], "sim_errors": c[2], "identity": np.int64(c[4]).view(np.float64), "status": c[5]}
            related_all = ((np.arange(n)[:, None] % K) == (np.arange(n)[None, :] % K))[~np.eye(n, dtype=bool)]
            out["cpu_baseline"] = anim_cpu_baseline(args, data, n, related_all, gpu_lookup)
            out["cpu_baseline"]["speedup_gpu_over_cpu_job"] = out["value"] / out["cpu_baseline"]["value"]
        if world == 1 and not bare:
            out["related_only"] = related_only_record(eng, args, stages)
            out["unrelated_only"] = unrelated_only_record(eng, args, ids)
            out["sketch_mode"] = _side_record(sketch_record, eng, ids, n, K, dense, covered)
            cb = out.get("cpu_baseline")
            if cb and cb.get("cpu_s_per_related_pair"):
                # the two halves of the job priced separately: a genus-level job is all related pairs
                cb["speedup_related_only"] = out["related_only"]["pairs_per_s"] / (cb["cores"] / cb["cpu_s_per_related_pair"])
                if out["related_only"].get("pipelined_calls"):
                    cb["speedup_related_only_pipelined_calls"] = out["related_only"]["pipelined_calls"]["pairs_per_s"] / (cb["cores"] / cb["cpu_s_per_related_pair"])
                if out["related_only"].get("steady"):
                    cb["speedup_related_only_steady"] = out["related_only"]["steady"]["pairs_per_s"] / (cb["cores"] / cb["cpu_s_per_related_pair"])
                if out["unrelated_only"] and cb.get("cpu_s_per_unrelated_pair"):
                    cb["speedup_unrelated_only"] = out["unrelated_only"]["pairs_per_s"] / (cb["cores"] / cb["cpu_s_per_unrelated_pair"])
                cb["note_speedups"] = ("the GPU's all-related family job (related_only) and all-unrelated job (unrelated_only), each ONE call, against host "
                                       "threads / CPU seconds per pair of that kind; speedup_gpu_over_cpu_job is the C4 mix (97.6 % unrelated)")
        if world == 1 and not args.no_tetra:
            out["tetra"] = tetra_subrecord(eng, local, args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
        failed = []
        if sha_check and not sha_check["ok"]:
            failed.append(f"result grid sha1 {sha} != expected {sha_check['expected']}")
        if parity_indep and parity_indep.get("checked") and parity_indep["identical"] != parity_indep["checked"]:
            failed.append(f"independent-oracle parity {parity_indep['identical']}/{parity_indep['checked']}: {parity_indep['first_difference']}")
        if failed:
            print("bench.py: RESULT CHECK FAILED: " + "; ".join(failed), file=sys.stderr, flush=True)
            eng.close()
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


# =====================================================================================================================
# ANIb fragment mode (C5)
# =====================================================================================================================
def c5_length(g):
    """SURVEY.md §8(d) set C5: L_g = 1 000 000 + (g * 22 045 mod 11 000 001)  (1-12 Mb)."""
    return 1_000_000 + (g * 22_045) % 11_000_001


def anib_traffic(kernel, args, n):
    """HBM bytes per launch of the dominant C5 kernel from the committed PMC passes (profiles/pmc_anib.json: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of `--workload anib --steps 1 --warmup 1`, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md) — only for the configuration they were taken on."""
    pmc = ROOT / "profiles" / "pmc_anib.json"
    if not pmc.exists() or (n, args.seed) != (500, 20250302):
        return None
    return json.loads(pmc.read_text()).get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch")


def run_anib(args, rank, world, local, dist, torch):
    """C5: N = 500 synthetic genomes of 1-12 Mb, pyani's ANIb on the engine's fragment mode: the 1020-nt fragments of every
    genome against every other genome.  A step = `--rows-per-step` (10) fragmented genomes x all N - 1 subjects; N > 1 = strong
    scaling of the same steps (rows dealt over the ranks, one all-gather per step), as for ANIm."""
    from pyani_amd import _lib, parallel, synth
    from pyani_amd.engine import Engine
    eng = Engine(local)
    n, R = args.genomes, max(1, min(args.rows_per_step, args.genomes))
    K = (n + 24) // 25
    t_prep = time.perf_counter()
    with ThreadPoolExecutor(max(1, min(64, (os.cpu_count() or 2) // max(1, world)))) as ex:
        data = list(ex.map(lambda g: synth.genome(args.seed, n, g, c5_length(g)), range(n)))
    ids_np = np.asarray([eng.add_genome(s_, o_) for s_, o_ in data], dtype=np.int32)
    eng.upload()
    t_prep = time.perf_counter() - t_prep
    dev = torch.device("cuda", local)
    lens = np.array([len(d[0]) for d in data], dtype=np.int64)

    def compute(pairs):   # pairs: (fragmented genome, subject genome)
        return parallel.anib_records_to_tensor(eng.anib_pairs(ids_np[pairs[:, 0]], ids_np[pairs[:, 1]]), dev)

    tiles = {}

    def step(k, keep=False):
        rows = [(k * R + i) % n for i in range(R)]
        if dist is not None:
            grid = parallel.anim_allgather(compute, n, dev, rows=rows)
        else:
            pairs = parallel.anim_pair_array(n, rows)
            grid = torch.zeros((len(rows), n, parallel.ANIM_FIELDS), dtype=torch.int64, device=dev)
            grid[torch.from_numpy(np.repeat(np.arange(len(rows)), n - 1)).to(dev), torch.from_numpy(pairs[:, 1]).to(dev)] = compute(pairs)
        if keep:
            tiles[k] = (rows, grid)

    def fence():
        eng.sync()
        if not REHEARSAL:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not REHEARSAL:
            torch.cuda.synchronize()

    t_cold = time.perf_counter()
    for k in range(args.warmup):
        step(k)
        if k == 0:
            fence()
            t_cold = time.perf_counter() - t_cold      # the first step also builds the seed lists of every genome it touches
    fence()
    if args.warmup == 0:
        t_cold = None
    stages = [_lib.K_ANIM_SEED, _lib.K_ANIM_HIT, _lib.K_ANIB_BUCKET, _lib.K_ANIB_FRAG]
    eng.profile_reset()
    eng.profile_config(kernel_mask=sum(1 << s_ for s_ in stages), every_n=1)
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        step(k, keep=True)
    fence()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = {eng.kernel_name(s_): eng.profile_get(s_) for s_ in stages}
    if rank == 0:
        done_rows = [q for k in tiles for q in tiles[k][0]]
        rowv = np.array(done_rows)
        g = torch.cat([tiles[k][1] for k in sorted(tiles)]).cpu().numpy()
        offdiag = rowv[:, None] != np.arange(n)[None, :]
        related_m = ((rowv[:, None] % K) == (np.arange(n)[None, :] % K)) & offdiag
        pid = g[:, :, 4].view(np.float64)
        kept, nfr = g[:, :, 3], g[:, :, 2]
        pairs_done = int(offdiag.sum())
        frags_done = int(nfr[offdiag].sum())
        alg_bytes = float(sum(((lens[q] + 3) // 4 + (lens + 3) // 4 + 32).sum() - ((lens[q] + 3) // 4 * 2 + 32) for q in done_rows))
        dom = max(prof, key=lambda name: prof[name][0])
        dom_ms, dom_n = prof[dom]
        achieved = alg_bytes / world / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
        out = {
            "metric": "genome-pairs/sec (ordered pairs) for the N x N ANIb matrices: 1020-nt fragments of one genome searched in the "
                      "other (blastn -task blastn equivalent) + parse_blast_tab, genomes resident in HBM",
            "value": pairs_done / elapsed, "unit": "genome-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32 DP scores and counts, f64 mean identity", "data": "synthetic",
            "related_pairs_per_s": int(related_m.sum()) / elapsed, "fragments_per_s": frags_done / elapsed,
            "config": {
                "workload": f"C5: ANIb fragment mode on {n} synthetic genomes of 1-12 Mb (SURVEY.md §8(d): L_g = 1 000 000 + (g * 22 045 mod "
                            f"11 000 001), seed {args.seed}), fragment size 1020; a step = {R} fragmented genomes x all {n - 1} subjects",
                "genomes": n, "rows_per_step": R, "pairs_timed": pairs_done, "related_pairs_timed": int(related_m.sum()),
                "fragments_timed": frags_done, "grid_pairs": n * (n - 1), "wall_s_grid": elapsed / pairs_done * n * (n - 1),
                "related_pairs_with_hits": int(((kept > 0) & related_m).sum()), "unrelated_pairs_with_hits": int(((kept > 0) & ~related_m & offdiag).sum()),
                "identity_related_min_med_max": [float(x) for x in np.percentile(pid[(kept > 0) & related_m], [0, 50, 100])]
                if ((kept > 0) & related_m).any() else None,
                "coverage_related_median": float(np.median((g[:, :, 0] / lens[rowv][:, None])[(kept > 0) & related_m]))
                if ((kept > 0) & related_m).any() else None,
                "parallelism": f"1 process/GPU x {world}; genomes replicated; each step's rows dealt over the ranks; one RCCL "
                               f"all-gather per step" if world > 1 else "1 GPU",
                "host_prep_s": t_prep,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": anib_traffic(dom, args, n), "launches": int(dom_n), "avg_launch_ms": dom_ms / max(dom_n, 1),
                "definition": "SURVEY.md §8(d): ceil(Lq/4) + ceil(Ls/4) + 32 B per ordered pair / the HIP-event time of the stage that took "
                              "longest; the fragment DP is LDS / VALU work, not an HBM stream: the fraction is small by construction",
                "stage_ms": {name: round(ms, 3) for name, (ms, _) in prof.items()}, "timed_region_ms": round(elapsed * 1e3, 3),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, str(ROOT / "oracle"))
            import anib_cpu
            threads = os.cpu_count() or 1
            k = args.cpu_pairs or min(threads, 32)
            rng = np.random.RandomState(777)
            rel = [(int(q), int((q + K * int(rng.randint(1, max(2, n // K)))) % n)) for q in rng.randint(0, n, size=k // 2)]
            rel = [(q, s_) for q, s_ in rel if q != s_ and q % K == s_ % K]
            unrel = [(int(q), int(s_)) for q, s_ in zip(rng.randint(0, n, size=k), rng.randint(0, n, size=k)) if q % K != s_ % K][: k - len(rel)]
            sample = rel + unrel
            t0 = time.perf_counter()
            with ThreadPoolExecutor(min(threads, len(sample))) as ex:     # ctypes releases the GIL: one pair per thread
                def one(p):
                    t1 = time.perf_counter()
                    rows = anib_cpu.anib_cpu_pair(data[p[0]], data[p[1]])
                    return time.perf_counter() - t1, anib_cpu.reduce_rows(rows)[:3]
                got = list(ex.map(one, sample))
            wall = time.perf_counter() - t0
            t_rel = float(np.mean([x[0] for x in got[: len(rel)]])) if rel else 0.0
            t_unrel = float(np.mean([x[0] for x in got[len(rel):]])) if unrel else 0.0
            n_rel_job = n * (n // K - 1)
            job_cpu = n_rel_job * t_rel + (n * (n - 1) - n_rel_job) * t_unrel
            out["cpu_baseline"] = {
                "value": n * (n - 1) / (job_cpu / threads), "unit": "genome-pairs/s", "cores": threads, "kind": "port",
                "variant": "own-cpu (oracle/anib_cpu.cpp: host build of the fragment-mode statement; NOT BLAST+)",
                "sample": f"blastn is not on this box; {len(sample)} ordered pairs ({len(rel)} related, {len(unrel)} unrelated) one per host thread "
                          f"({wall:.1f} s wall): {t_rel:.1f} s per related, {t_unrel:.1f} s per unrelated pair; extrapolated linearly to the N x N job "
                          f"on {threads} threads",
                "job_seconds_extrapolated": job_cpu / threads,
            }
            out["cpu_baseline"]["speedup_gpu_over_cpu_job"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


# =====================================================================================================================
# TETRA (C2)
# =====================================================================================================================
def tetra_cpu_baseline(eng_z, sample, n_genomes, n_pairs):
    """Time the pure-Python port of pyani's TETRA (oracle/tetra_port.py — checker code, used here ONLY as the
    reported CPU baseline) on a bounded sample and extrapolate to the whole job: pyani runs TETRA sequentially on
    one core (scripts/average_nucleotide_identity.py:606-608), so the job time is n*t_genome + pairs*t_pair."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import tetra_port
    from pyani_amd.tetra import TETRAMERS
    t_gen, checked = [], 0
    for k, (seq, off) in enumerate(sample):
        recs = [bytes(seq[int(off[r]):int(off[r + 1])]).decode("latin-1") for r in range(len(off) - 1)]
        t0 = time.perf_counter()
        z = tetra_port.zscores_from_counts(*tetra_port.count_kmers(recs))
        t_gen.append(time.perf_counter() - t0)
        # full-size parity check while we are here: the port's Z == the GPU's Z, bit for bit
        gpu = {TETRAMERS[t]: float(eng_z[k, t]) for t in range(256)}
        checked += int(all(gpu[t] == v for t, v in z.items()) and len(z) == 256)
    # correlation cost: 40 genomes' worth of GPU Z vectors -> 780 pairs
    m = min(40, eng_z.shape[0])
    zdicts = {f"o{k:03d}": {TETRAMERS[t]: float(eng_z[k, t]) for t in range(256)} for k in range(m)}
    t0 = time.perf_counter()
    tetra_port.correlations(zdicts)
    t_pair = (time.perf_counter() - t0) / (m * (m - 1) / 2)
    t_genome = sum(t_gen) / len(t_gen)
    total = n_genomes * t_genome + n_pairs * t_pair
    return {
        "value": n_pairs / total, "unit": "genome-pairs/s", "cores": 1, "kind": "port",
        "sample": f"{len(sample)} of {n_genomes} genomes counted+Z-scored by oracle/tetra_port.py "
                  f"({t_genome:.2f} s/genome), {m * (m - 1) // 2} pairs correlated ({t_pair * 1e3:.3f} ms/pair); "
                  f"extrapolated linearly to {n_genomes} genomes + {n_pairs} pairs = {total:.0f} s (pyani runs TETRA on 1 core; "
                  f"port vs the imported reference on one 4.02 Mb genome: see BASELINE.md)",
        "job_seconds_extrapolated": total, "gpu_parity_on_sample": f"{checked}/{len(sample)} genomes bit-identical",
    }


def run_tetra(args, rank, world, local, dist, torch):
    from pyani_amd import _lib
    from pyani_amd.engine import Engine
    eng = Engine(local)
    n_local, n_total = args.genomes, args.genomes * world
    g0 = rank * n_local

    # ---- synthetic inputs -> HBM (outside the timed region) ----------------------------------------------------
    t_prep = time.perf_counter()
    data = synth_genomes(args.seed, n_total, args.length, g0, g0 + n_local, world)
    ids = [eng.add_genome(s_, o_) for s_, o_ in data]
    eng.upload()
    ids_arr = np.ascontiguousarray(ids, dtype=np.int32)
    t_prep = time.perf_counter() - t_prep
    alg_bytes, bases = eng.tetra_algorithmic_bytes(ids_arr.tolist())

    ag = None
    if world > 1:
        from pyani_amd import parallel
        ag = parallel.TetraAllGather(n_total, torch.device("cuda", local))
        assert (ag.lo, ag.hi) == (g0, g0 + n_local)
        ids_list = ids_arr.tolist()

        def compute_z(z_loc, p_loc):
            eng.tetra_zscores_dev(ids_list, z_loc.data_ptr(), p_loc.data_ptr())

        def compute_rows(z_all, p_all, lo, nrows, rows):
            eng.tetra_corr_rows_dev(z_all.data_ptr(), p_all.data_ptr(), n_total, lo, nrows, rows.data_ptr())

    def step():
        if world == 1:
            eng.tetra_matrix_enqueue(ids_arr, fetch_z=False)   # the product of a pass is the N x N matrix
        else:
            ag.run(compute_z, compute_rows)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_reset()
    # HIP events around the count kernel only, on every 4th launch: measuring inside the timed region must not
    # perturb it (an event pair costs ~20 us of stream bubbles; see profiles/).
    eng.profile_config(kernel_mask=1 << _lib.K_TETRA_COUNT, every_n=4)
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    count_ms, count_n = eng.profile_get(_lib.K_TETRA_COUNT)
    if rank == 0:
        if world == 1:
            eng.tetra_matrix_enqueue(ids_arr, fetch_z=True)   # untimed: also bring Z back for the checks below
            z, present, corr = eng.tetra_matrix_fetch(n_local)
        else:
            z, corr = ag.z_all.cpu().numpy(), ag.corr.cpu().numpy()
        assert np.isfinite(corr).all() and (np.diag(corr) == 1.0).all() and (corr == corr.T).all()
        pairs = n_total * (n_total - 1) // 2
        ms_step = elapsed / args.steps * 1e3
        avg_count_s = count_ms / max(count_n, 1) * 1e-3
        achieved = alg_bytes / avg_count_s / 1e9 if count_n else 0.0
        traffic = None
        pmc = ROOT / "profiles" / "pmc_tetra_count.json"
        if pmc.exists() and world == 1 and (args.genomes, args.length, args.seed) == (200, 5_000_000, 20250228):
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
        out = {
            "metric": "genome-pairs/sec for the TETRA N x N matrix (counts + Z + Pearson), inputs resident in HBM",
            "value": pairs / (elapsed / args.steps), "unit": "genome-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 counts + f64 Z/Pearson", "data": "synthetic",
            "config": {
                "workload": f"C2: TETRA on {n_total} synthetic ~{args.length / 1e6:g} Mb genomes "
                            f"(SURVEY.md §8(d) generator, seed {args.seed}), {pairs} unordered pairs",
                "genomes_per_gpu": n_local, "genomes": n_total, "bases_per_gpu": int(bases), "pairs": pairs,
                "parallelism": "1 process/GPU; genomes sharded; RCCL all-gather of Z then of matrix rows" if world > 1 else "1 GPU",
                "wall_s_matrix": elapsed / args.steps, "genomes_per_s": n_total / (elapsed / args.steps),
                "host_prep_s": t_prep,
            },
            "roofline": {
                "bound": "hbm", "kernel": "tetra_count_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_us": avg_count_s * 1e6, "launches": int(count_n),
                "frac_of_measured_copy_peak_6290": achieved / 6290.0,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = tetra_cpu_baseline(z, data[: args.cpu_genomes], n_total, pairs)
            out["cpu_baseline"]["speedup_gpu_over_cpu_job"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def main():
    args = parse_args()
    # The engine's two workers own one HIP stream each and rely on their kernels overlapping.  ROCm maps the streams of a process onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL has added its own streams the two workers can land on ONE queue and
    # serialise: measured on MI355X with one rank through the real backend, 53.7 k pairs/s against 58.5 k plain, 56.8 k with 8 queues
    # (16: the same; the plain run does not move).  Must be in the environment before the HIP runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (see module docstring)")
        args.gpus = world
    import torch
    if REHEARSAL:      # (tests: the N > 1 control flow on CPU over gloo with a stub engine; never a measurement)
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        if args.workload != "anim":
            sys.exit("the rehearsal covers the anim workload")
        args.no_cpu_baseline = args.no_tetra = True
        return run_anim(args, rank, world, local, dist, torch)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the engine has no CPU fallback")
    # debugging aid for 1-GPU boxes: run the N>1 code path with every rank on GPU 0 over gloo (never the default)
    one_gpu_debug = os.environ.get("PYANI_BENCH_DEBUG_ONE_GPU") == "1"
    if one_gpu_debug:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    # PYANI_BENCH_FORCE_DIST=1 (tests): take the collective path with ONE rank too, so that the RCCL calls themselves run on a
    # 1-GPU box (launch through torch.distributed.run --nproc-per-node 1)
    if world > 1 or os.environ.get("PYANI_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if one_gpu_debug:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.workload == "anim":
        return run_anim(args, rank, world, local, dist, torch)
    if args.workload == "anib":
        return run_anib(args, rank, world, local, dist, torch)
    return run_tetra(args, rank, world, local, dist, torch)


if __name__ == "__main__":
    main()
