#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: TETRA N x N matrix on synthetic ~5 Mb genomes (BASELINE.json
configs[1], SURVEY.md §8(d) set C2), genome-pairs/s on MI355X, with the kernel roofline and the CPU baseline.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one full pass of the path over the batch, inputs already resident in HBM (2-bit codes + 1-bit mask):
  N = 1 : count kernel -> finalize/Z -> stats -> Pearson -> D2H of Z and the matrix       (200 genomes, 19 900 pairs)
  N > 1 : weak scaling, 200 genomes per GPU (job = 200*N genomes, all 200N(200N-1)/2 pairs): every rank counts its
          genomes, RCCL all-gather of Z (N*200 x 256 f64 + presence), every rank computes its row block of the
          matrix, RCCL all-gather of the rows.  (two collectives, both tiny.)
Rank 0 prints ONE JSON line.

  --workload anim : the ANIm side of the same metric on the same genomes (C3: all 39 800 ordered pairs; a step is one
          pass over the whole grid; N > 1 = strong scaling, pair grid dealt by reference row, one all-gather).  Not the
          default: BASELINE.json quotes the metric on configs[1] (TETRA) for N = 1.
"""
import argparse
import json
import os
import sys
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
    ap.add_argument("--workload", choices=["tetra", "anim"], default="tetra",
                    help="tetra = C2 (the default, BASELINE.json configs[1]); anim = C3 (all ordered pairs of the same genomes)")
    ap.add_argument("--steps", type=int, default=None, help="default 50 (tetra) / 3 (anim)")
    ap.add_argument("--warmup", type=int, default=None, help="default 5 (tetra) / 1 (anim)")
    ap.add_argument("--genomes", type=int, default=200, help="genomes per GPU (C2: 200)")
    ap.add_argument("--length", type=int, default=5_000_000, help="ancestor length in bases (C2: 5 Mb)")
    ap.add_argument("--seed", type=int, default=20250228)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-genomes", type=int, default=2, help="genomes timed by the CPU baseline leg")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 50 if args.workload == "tetra" else 3
    if args.warmup is None:
        args.warmup = 5 if args.workload == "tetra" else 1
    return args


def run_anim(args, rank, world, local, dist, torch):
    """C3-shaped ANIm workload: every ordered pair of `--genomes` synthetic genomes (replicated on every GPU).  A step is
    one pass over the whole ordered-pair grid; with N GPUs the grid is dealt by reference row and assembled with ONE
    all-gather (pyani_amd/parallel.py), i.e. STRONG scaling of a fixed job."""
    from pyani_amd import parallel, synth
    from pyani_amd.engine import Engine
    eng = Engine(local)
    n = args.genomes
    t_prep = time.perf_counter()
    with ThreadPoolExecutor(max(1, min(16, (os.cpu_count() or 2) // max(1, world)))) as ex:
        data = list(ex.map(lambda g: synth.genome(args.seed, n, g, args.length), range(n)))
    ids = [eng.add_genome(s_, o_) for s_, o_ in data]
    eng.upload()
    t_prep = time.perf_counter() - t_prep
    dev = torch.device("cuda", local)
    mine = parallel.anim_pair_shard(n, rank, world)

    def compute(pairs):
        return parallel.anim_records_to_tensor(eng.anim_pairs([ids[q] for q, _ in pairs], [ids[s_] for _, s_ in pairs]), dev)

    grid = None

    def step():
        nonlocal grid
        grid = parallel.anim_allgather(compute, n, dev) if world > 1 else compute(mine)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        pairs = n * (n - 1)
        g = grid.cpu().numpy().reshape(-1, parallel.ANIM_FIELDS)
        status = g[:, 5] if world == 1 else g[np.arange(n * n) % (n + 1) != 0, 5]
        step_s = elapsed / args.steps
        lens = [len(d[0]) for d in data]
        alg_bytes = sum((lens[q] + 3) // 4 + (lens[s_] + 3) // 4 + 32 for q in range(n) for s_ in range(n) if q != s_)
        out = {
            "metric": "genome-pairs/sec (ordered pairs) for the ANIm N x N grid: nucmer --mum + delta-filter -1 + parse_delta "
                      "equivalent, genomes resident in HBM",
            "value": pairs / step_s, "unit": "genome-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32 DP keys (score << 15 | errors), f64 identity", "data": "synthetic",
            "config": {
                "workload": f"C3: ANIm on {n} synthetic ~{args.length / 1e6:g} Mb genomes (SURVEY.md §8(d) generator, seed "
                            f"{args.seed}), all {pairs} ordered pairs",
                "genomes": n, "pairs": pairs, "pairs_with_alignment": int((status == 0).sum()),
                "parallelism": "1 process/GPU; genomes replicated; pair grid dealt by reference row; one RCCL all-gather"
                               if world > 1 else "1 GPU",
                "wall_s_grid": step_s, "host_prep_s": t_prep,
            },
            "roofline": {
                "bound": "hbm", "kernel": "whole ANIm pipeline (no single HBM-bound kernel: the extension DP is VALU-issue bound, "
                                          "the seed stage streams 16 MB of k-mer list per pair; see DESIGN.md §8)",
                "achieved": alg_bytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": int(alg_bytes),
            },
            "cpu_baseline": {"value": None, "unit": "genome-pairs/s", "cores": 0, "kind": "reference",
                             "sample": "unavailable: nucmer / delta-filter (MUMmer 3.23) are absent from this image and from "
                                       "the GPU box, and their source is not in the reference tree (SURVEY.md §8c)"},
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def cpu_baseline(eng_z, sample, n_genomes, n_pairs):
    """Time the pure-Python port of pyani's TETRA (oracle/tetra_port.py — checker code, used here ONLY as the
    reported CPU baseline) on a bounded sample and extrapolate to the whole job: pyani runs TETRA sequentially on
    one core (scripts/average_nucleotide_identity.py:606-608), so the job time is n*t_genome + pairs*t_pair."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import tetra_port
    from pyani_amd.tetra import TETRAMERS
    t_gen, checked = [], 0
    zs = {}
    for k, (seq, off) in enumerate(sample):
        recs = [bytes(seq[int(off[r]):int(off[r + 1])]).decode("latin-1") for r in range(len(off) - 1)]
        t0 = time.perf_counter()
        z = tetra_port.zscores_from_counts(*tetra_port.count_kmers(recs))
        t_gen.append(time.perf_counter() - t0)
        zs[f"g{k}"] = z
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
                  f"extrapolated linearly to {n_genomes} genomes + {n_pairs} pairs = {total:.0f} s (pyani runs TETRA on 1 core)",
        "job_seconds_extrapolated": total, "gpu_parity_on_sample": f"{checked}/{len(sample)} genomes bit-identical",
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (see module docstring)")
        args.gpus = world
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the engine has no CPU fallback")
    # debugging aid for 1-GPU boxes: run the N>1 code path with every rank on GPU 0 over gloo (never the default)
    one_gpu_debug = os.environ.get("PYANI_BENCH_DEBUG_ONE_GPU") == "1"
    if one_gpu_debug:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_gpu_debug:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    if args.workload == "anim":
        return run_anim(args, rank, world, local, dist, torch)
    from pyani_amd import _lib, synth
    from pyani_amd.engine import Engine
    eng = Engine(local)
    n_local, n_total = args.genomes, args.genomes * world
    g0 = rank * n_local

    # ---- synthetic inputs -> HBM (outside the timed region) ----------------------------------------------------
    t_prep = time.perf_counter()
    with ThreadPoolExecutor(max(1, min(16, (os.cpu_count() or 2) // max(1, world)))) as ex:
        data = list(ex.map(lambda g: synth.genome(args.seed, n_total, g, args.length), range(g0, g0 + n_local)))
        ids = list(ex.map(lambda d: eng.add_genome(d[0], d[1]), data))
    eng.upload()
    order = np.argsort(ids)
    data = [data[k] for k in order]
    ids_arr = np.ascontiguousarray(sorted(ids), dtype=np.int32)
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

    prof = {eng.kernel_name(k): eng.profile_get(k) for k in range(4)}
    count_ms, count_n = prof["tetra_count_kernel"]
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
            out["cpu_baseline"] = cpu_baseline(z, data[: args.cpu_genomes], n_total, pairs)
            out["cpu_baseline"]["speedup_gpu_over_cpu_job"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
