#!/usr/bin/env python3
"""gpurun_out/<round> (scratch; default r06) -> profiles/ (tracked): the rocprofv3 summaries of `bench.py --gpus 1 --steps 1 --warmup 1
--no-cpu-baseline --no-tetra` (C4, one step = a tenth of the grid) that bench.py's roofline block reads.

  profiles/<round>_anim_C4_rocprofv3_kernel_stats_one_worker.csv / ..._two_workers.csv   --kernel-trace --stats
  profiles/<round>_anim_C4_pmc_fetch_summary.csv / _write_summary.csv / _sq_summary.csv     --pmc passes (one worker), per kernel
  profiles/pmc_anim.json   per bench stage: HBM bytes per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md §HBM);
                           extension_valu_per_cell = SQ_INSTS_VALU of the extension kernels / the engines' own DP-cell count of
                           the same command (PYANI_PN_STATS run); end_to_end_cold_s_measured from the --cold-e2e run if present
Usage: python tools/summarize_round_profiles.py [r06]"""
import csv
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"      # the round whose scratch directory (gpurun_out/<round>) is summarised
SRC, DST = ROOT / "gpurun_out" / RND, ROOT / "profiles"
STAGE = {   # rocprofv3 kernel -> the stage name bench.py reports (pg_kernel_name)
    "anim_seed_kernel": "anim_seed_kernel",
    "anim_hit_kernel": "anim_hit_kernels", "anim_hit_scatter_kernel": "anim_hit_kernels", "anim_hoff_kernel": "anim_hit_kernels", "anim_scatter_kernel": "anim_hit_kernels",
    "anim_cluster_wave_kernel": "anim_cluster_wave_kernel", "anim_cluster_prep_kernel": "anim_cluster_wave_kernel",
    "anim_chain_range_kernel": "anim_cluster_wave_kernel", "anim_chain_merge_kernel": "anim_cluster_wave_kernel",
    "anim_postnuc_gaplist_kernel": "anim_postnuc_gap_kernels", "anim_postnuc_gapbig_kernel": "anim_postnuc_gap_kernels",
    "anim_postnuc_gaplane_kernel<16>": "anim_postnuc_gap_kernels", "anim_postnuc_gaplane_kernel<32>": "anim_postnuc_gap_kernels",
    "anim_postnuc_gaplane_kernel<59>": "anim_postnuc_gap_kernels",
    "anim_postnuc_fwd_kernel": "anim_postnuc_fwd_kernel",
    "anim_postnuc_rehearse_kernel": "anim_postnuc_rehearse_kernel+anim_postnuc_bwd_kernel", "anim_postnuc_bwd_kernel": "anim_postnuc_rehearse_kernel+anim_postnuc_bwd_kernel",
    "anim_postnuc_kernel": "anim_postnuc_kernel",
    "anim_postnuc_forced_kernel": "anim_postnuc_forced_kernels", "anim_postnuc_forced_wide_kernel": "anim_postnuc_forced_kernels",
    "anim_postnuc_forced_huge_kernel": "anim_postnuc_forced_kernels", "anim_postnuc_forced_strips_kernel": "anim_postnuc_forced_kernels",
    "anim_finish_kernel": "anim_finish_kernel",
}
EXT = [k for k in STAGE if k.startswith("anim_postnuc_")]


def rows(path):
    return {r["kernel"]: r for r in csv.DictReader(open(path))} if path.exists() else {}


for w, name in (("kt1", "one_worker"), ("kt2", "two_workers")):
    f = SRC / f"{w}_kernel_stats.csv"
    if f.exists():
        shutil.copyfile(f, DST / f"{RND}_anim_C4_rocprofv3_kernel_stats_{name}.csv")
for w in ("fetch", "write"):
    f = SRC / f"pmc_{w}_summary.csv"
    if f.exists():
        shutil.copyfile(f, DST / f"{RND}_anim_C4_pmc_{w}_summary.csv")
if (SRC / "sq_summary.csv").exists():
    shutil.copyfile(SRC / "sq_summary.csv", DST / f"{RND}_anim_C4_pmc_sq_summary.csv")
fetch, write, sq = rows(SRC / "pmc_fetch_summary.csv"), rows(SRC / "pmc_write_summary.csv"), rows(SRC / "sq_summary.csv")
out = {"round": RND,
       "command": "python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-tetra under rocprofv3 --pmc <counter> with PYANI_ANIM_WORKERS=1 "
                  "(C4, a step = a tenth of the grid: 99 900 ordered pairs; 12 launches per kernel: the warm-up step, the timed one and the one-worker roofline pass over all ten tiles)",
       "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md §HBM; exact for wide coalesced streams, "
                     "an upper bound for narrower accesses); WRITE_SIZE as reported (uncalibrated)"}
stage_bytes = {}
for k, st in STAGE.items():
    f, w = fetch.get(k), write.get(k)
    if f and w:
        n = max(1, int(f["launches"]))
        b = 2 * 1024 * float(f["FETCH_SIZE_sum"]) / n + 1024 * float(w["WRITE_SIZE_sum"]) / max(1, int(w["launches"]))
        stage_bytes.setdefault(st, {"kernels": {}, "hbm_bytes_per_launch": 0})
        stage_bytes[st]["kernels"][k] = {"FETCH_SIZE_KiB_per_launch": float(f["FETCH_SIZE_sum"]) / n, "WRITE_SIZE_KiB_per_launch": float(w["WRITE_SIZE_sum"]) / max(1, int(w["launches"]))}
        stage_bytes[st]["hbm_bytes_per_launch"] += int(b)
out.update(stage_bytes)
# instructions per DP cell of the extension stage
stats = SRC / "bench_c4_stats.err"
cells = None
if stats.exists():
    # one "[pn-stats] units ..." line per launch of the SAME command as the PMC passes (3 launches: warm-up, timed, roofline step):
    # averages per launch on both sides of the ratio
    ms = re.findall(r"regs: calls \d+ steps (\d+) cells (\d+).*?lds: calls \d+ steps \d+ cells (\d+).*?global: calls \d+ steps \d+ cells (\d+)", stats.read_text())
    if ms:
        cells = sum(int(m[1]) + int(m[2]) + int(m[3]) for m in ms) / len(ms)
        out["extension_cells_per_launch"] = cells
        out["extension_anti_diagonals_per_launch"] = sum(int(m[0]) for m in ms) / len(ms)
        out["extension_launches_counted"] = len(ms)
        (DST / f"{RND}_pn_stats_C4_step.txt").write_text("".join(l + "\n" for l in stats.read_text().splitlines() if "pn-stats" in l))
# per kernel class (bench.py: roofline.valu_issue.per_kernel): SQ_INSTS_VALU summed over the class's kernels and ALL launches of the SQ pass /
# the engines' cells of that class summed over ALL launches of the stats run of the same command
if stats.exists() and sq:
    CLASS = {"gaps": ["anim_postnuc_gapbig_kernel"], "forward": ["anim_postnuc_fwd_kernel"], "backward_ahead": ["anim_postnuc_bwd_kernel"],
             "walks": ["anim_postnuc_kernel"], "forced_narrow": ["anim_postnuc_forced_kernel"], "forced_wide": ["anim_postnuc_forced_wide_kernel"],
             "forced_group": ["anim_postnuc_forced_huge_kernel"]}
    LABEL = {"gaps": "gaps", "forward": "forward", "backward_ahead": "backward-ahead", "walks": "walks", "forced_narrow": "forced narrow",
             "forced_512_1024": "forced 512-1024", "forced_2048": "forced 2048", "forced_group": "forced group 8192"}
    txt = stats.read_text()
    def class_cells(label):
        return sum(int(m) for m in re.findall(r"diagonal engine in " + re.escape(label) + r"\s*: \d+ calls, \d+ anti-diagonals, (\d+) cells", txt))
    by_class, by_class_salu = {}, {}
    for cls, kernels in CLASS.items():
        labels = ["forced 512-1024", "forced 2048"] if cls == "forced_wide" else [LABEL[cls]]
        c = sum(class_cells(l) for l in labels)
        v = sum(float(sq[k]["SQ_INSTS_VALU_sum"]) for k in kernels if k in sq)
        sa = sum(float(sq[k]["SQ_INSTS_SALU_sum"]) for k in kernels if k in sq)
        if c and v:
            for name in (["forced_512_1024", "forced_2048"] if cls == "forced_wide" else [cls]):
                by_class[name] = v / c
                by_class_salu[name] = sa / c
    out["valu_per_cell_by_class"] = by_class
    out["salu_per_cell_by_class"] = by_class_salu
if cells and sq:
    valu = sum(float(sq[k]["SQ_INSTS_VALU_sum"]) / max(1, int(sq[k]["launches"])) for k in EXT if k in sq)
    salu = sum(float(sq[k]["SQ_INSTS_SALU_sum"]) / max(1, int(sq[k]["launches"])) for k in EXT if k in sq)
    out["extension_valu_per_cell"] = valu / cells
    out["extension_salu_per_cell"] = salu / cells
    out["extension_valu_source"] = ("rocprofv3 SQ_INSTS_VALU of the anim_postnuc_* kernels per launch (profiles/" + RND + "_anim_C4_pmc_sq_summary.csv) / the engines' "
                                    "DP-cell count of one launch of the same step (profiles/" + RND + "_pn_stats_C4_step.txt)")
cold = SRC / "cold_e2e.json"
if cold.exists():
    rec = json.loads([l for l in cold.read_text().splitlines() if l.startswith("{")][-1])
    out["end_to_end_cold_s_measured"] = rec["end_to_end_cold_s"]
    out["end_to_end_cold_breakdown_s"] = rec["config"]["seconds"]
    shutil.copyfile(cold, DST / f"{RND}_cold_e2e_C4.json")
(DST / "pmc_anim.json").write_text(json.dumps(out, indent=1) + "\n")
bench = SRC / "bench_n1.json"
if bench.exists() and bench.stat().st_size:
    shutil.copyfile(bench, DST / f"{RND}_bench_n1.json")
# ANIb (C5) at HEAD: the bench record, kernel trace and SQ pass of its steps
for src, dst in (("bench_anib_C5_n1.json", RND + "_bench_anib_C5_n1.json"), ("anib_kernel_stats.csv", RND + "_anib_C5_rocprofv3_kernel_stats.csv"),
                 ("anib_sq_summary.csv", RND + "_anib_C5_pmc_sq_summary.csv")):
    if (SRC / src).exists() and (SRC / src).stat().st_size:
        shutil.copyfile(SRC / src, DST / dst)
# ... and the HBM traffic of its kernels (FETCH_SIZE / WRITE_SIZE passes of one warm-up + one timed step): profiles/pmc_anib.json, read by bench.py
af, aw = rows(SRC / "anib_fetch_summary.csv"), rows(SRC / "anib_write_summary.csv")
if af and aw:
    for w in ("fetch", "write"):
        shutil.copyfile(SRC / f"anib_{w}_summary.csv", DST / f"{RND}_anib_C5_pmc_{w}_summary.csv")
    a = {"round": RND, "workload": "C5 (500 synthetic genomes of 1-12 Mb, seed 20250302), fragment mode",
         "command": "python bench.py --gpus 1 --workload anib --steps 1 --warmup 1 --no-cpu-baseline under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs)",
         "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md §HBM: exact for wide coalesced streams, an upper "
                       "bound for narrower accesses); WRITE_SIZE as reported (uncalibrated)", "kernels": {}}
    for k in af:
        if k in aw and (k.startswith("anib_") or k.startswith("anim_")):
            n = int(af[k]["launches"])
            fk, wk = float(af[k]["FETCH_SIZE_sum"]) / n, float(aw[k]["WRITE_SIZE_sum"]) / max(1, int(aw[k]["launches"]))
            a["kernels"][k] = {"launches": n, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk, "hbm_bytes_per_launch": int(2 * 1024 * fk + 1024 * wk)}
    (DST / "pmc_anib.json").write_text(json.dumps(a, indent=1) + "\n")
    print("anib:", {k: round(v["hbm_bytes_per_launch"] / 1e9, 2) for k, v in a["kernels"].items()})
# TETRA (C2) at HEAD: kernel trace + FETCH / WRITE / SQ passes of `bench.py --workload tetra --steps 20 --warmup 5 --no-cpu-baseline`
tk = SRC / "tetra_kernel_stats.csv"
if tk.exists():
    shutil.copyfile(tk, DST / f"{RND}_tetra_C2_rocprofv3_kernel_stats.csv")
    for w in ("fetch", "write", "sq"):
        if (SRC / f"tetra_{w}_summary.csv").exists():
            shutil.copyfile(SRC / f"tetra_{w}_summary.csv", DST / f"{RND}_tetra_C2_pmc_{w}_summary.csv")
    def count_row(path):
        return next((r for k, r in rows(path).items() if k.startswith("tetra_count_kernel")), None)
    f, w, q = count_row(SRC / "tetra_fetch_summary.csv"), count_row(SRC / "tetra_write_summary.csv"), count_row(SRC / "tetra_sq_summary.csv")
    kt = next((r for r in csv.DictReader(open(tk)) if "tetra_count_kernel" in r["Name"]), None)
    if f and w:
        fk, wk = float(f["FETCH_SIZE_sum"]) / int(f["launches"]), float(w["WRITE_SIZE_sum"]) / int(w["launches"])
        t = {"kernel": "tetra_count_kernel", "workload": "C2 (200 x 5 Mb synthetic genomes, seed 20250228)", "round": RND,
             "command": "python bench.py --gpus 1 --workload tetra --steps 20 --warmup 5 --no-cpu-baseline under rocprofv3 (--kernel-trace --stats; --pmc FETCH_SIZE; "
                        "--pmc WRITE_SIZE; --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES: separate runs)",
             "FETCH_SIZE_KiB_avg": fk, "WRITE_SIZE_KiB_avg": wk, "launches_sampled": int(f["launches"]),
             "correction": "gfx950 rocprofv3 tallies 128-B read requests at 64 B: FETCH_SIZE reads exactly half the bytes of a wide coalesced stream "
                           "(MI355X_MICROARCH.md §HBM) -> doubled. WRITE_SIZE taken as reported (uncalibrated, ~1 MB).",
             "hbm_bytes_per_launch": int(2 * 1024 * fk + 1024 * wk)}
        if kt:
            t["rocprofv3_avg_launch_us"] = float(kt["AverageNs"]) / 1e3
        if q:
            t["lds_bank_conflict_cycles_per_lds_active_cycle"] = float(q["SQ_LDS_BANK_CONFLICT_sum"]) / max(1.0, float(q["SQ_LDS_IDX_ACTIVE_sum"]))
            t["valu_instructions_per_launch"] = float(q["SQ_INSTS_VALU_sum"]) / int(q["launches"])
        (DST / "pmc_tetra_count.json").write_text(json.dumps(t, indent=1) + "\n")
        print("tetra:", json.dumps(t)[:600])
print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}, indent=1))
for st, v in stage_bytes.items():
    print(f"{st:60s} {v['hbm_bytes_per_launch'] / 1e9:9.2f} GB per launch")
