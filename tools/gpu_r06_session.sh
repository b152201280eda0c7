#!/bin/bash
# Round-6 GPU session steps (everything under gpurun_out/r06/).  Usage: tools/gpu_r06_session.sh step...
#   tests    the whole -m gpu suite            new      only the tests added this round
#   kt1      rocprofv3 kernel trace of ONE C4 step with ONE worker (launches not overlapped: clean per-kernel times)
#   kt2      the same with the shipping two workers
#   sq       one PMC pass (SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY)
#   fetch / write   FETCH_SIZE / WRITE_SIZE passes of the same command
#   stats4   one C4 step with the engines' own counters (PYANI_PN_STATS)
#   c4       short bench (4 steps)             bench    the driver's command
#   tetra    rocprofv3 of the TETRA workload at HEAD (kernel trace + FETCH/WRITE)     anib   C5 bench + kernel trace + SQ pass
R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
export PYANI_DEV_KNOBS=1
B1="python $R/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-tetra --no-side-records"      # (+ the one-worker roofline pass over all ten tiles: 12 launches per kernel)
for w in "$@"; do
case $w in
tests)
  timeout -k 10 1500 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -25 $O/pytest_gpu.log
  grep -q "pytest rc=0" $O/pytest_gpu.log || { echo "GPU tests failed: the remaining steps are skipped"; exit 1; } ;;
new)
  timeout -k 10 900 python -m pytest tests/test_anim_filter_oracle_gpu.py tests/test_anim_oracle_goldens_gpu.py -m gpu -q --timeout 400 --timeout-method=thread -x > $O/pytest_new.log 2>&1
  echo "pytest rc=$?" >> $O/pytest_new.log; tail -30 $O/pytest_new.log ;;
kt1|kt2)
  cd /tmp; rm -rf $O/$w
  W=1; [ $w = kt2 ] && W=2
  PYANI_ANIM_WORKERS=$W timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o kt -- $B1 > $O/${w}_bench.log 2>&1
  cd $R
  f=$(find $O/$w -name "*kernel_stats.csv" | head -1); cp $f $O/${w}_kernel_stats.csv; head -30 $f | cut -c1-220
  grep '^{' $O/${w}_bench.log | cut -c1-1800
  find $O/$w -name "*kernel_trace.csv" -size +20M -delete ;;
sq)
  cd /tmp; rm -rf $O/sq
  PYANI_ANIM_WORKERS=1 timeout -k 10 1200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o pmc -- $B1 > $O/sq_bench.log 2>&1
  cd $R
  python tools/summarize_pmc.py $O/sq $O/sq_summary.csv 2>&1 | tail -40
  find $O/sq -name "*counter_collection.csv" -size +20M -delete ;;
fetch|write)
  cd /tmp; rm -rf $O/pmc_$w
  C=FETCH_SIZE; [ $w = write ] && C=WRITE_SIZE
  PYANI_ANIM_WORKERS=1 timeout -k 10 1200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$w -o pmc -- $B1 > $O/pmc_$w.log 2>&1
  cd $R
  python tools/summarize_pmc.py $O/pmc_$w $O/pmc_${w}_summary.csv 2>&1 | tail -30
  find $O/pmc_$w -name "*counter_collection.csv" -size +20M -delete ;;
stats4)
  PYANI_PN_STATS=1 PYANI_ANIM_WORKERS=1 timeout -k 10 600 $B1 > $O/bench_c4_stats.log 2> $O/bench_c4_stats.err; echo "stats4 rc=$?"
  grep '^{' $O/bench_c4_stats.log | cut -c1-1500; grep "pn-stats" $O/bench_c4_stats.err | tail -16 ;;
grid)    # one whole grid timed (10 steps), no CPU leg: the quick comparison with the driver's value
  timeout -k 10 900 python bench.py --gpus 1 --steps 10 --warmup 2 --no-tetra --no-cpu-baseline --no-side-records > $O/bench_grid.log 2> $O/bench_grid.err; echo "grid rc=$?"
  grep '^{' $O/bench_grid.log > $O/bench_grid.json; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_grid.json").read())
r = d["roofline"]
print("value", round(d["value"]), "pairs/s, ms/step", round(d["ms_per_step"], 1), "sha", d["config"]["results_sha1_full_grid"])
print("one-worker stage sums over", r["tiles"], "tiles:", r["one_worker_step"]["stage_ms"])
for t in r["per_tile"]:
    print(t["tile"], t["kernel_ms_sum"], t["stage_ms"])
v = r["valu_issue"]
print("valu_issue", v and {k: v[k] for k in ("frac", "achieved", "cells", "extension_ms")})
for k, x in (v or {}).get("per_kernel", {}).items():
    print(" ", k, x)
PY
  ;;
c4)
  timeout -k 10 900 python bench.py --gpus 1 --steps 4 --warmup 1 --no-tetra --no-cpu-baseline > $O/bench_c4.log 2> $O/bench_c4.err; echo "c4 rc=$?"
  grep '^{' $O/bench_c4.log > $O/bench_c4.json; cut -c1-2500 $O/bench_c4.json; tail -5 $O/bench_c4.err ;;
bench)
  timeout 1500 python bench.py --gpus 1 --steps ${BENCH_STEPS:-20} --warmup ${BENCH_WARMUP:-5} > $O/bench_n1.log 2> $O/bench_n1.err; echo "bench rc=$?"
  grep '^{' $O/bench_n1.log > $O/bench_n1.json; cut -c1-3000 $O/bench_n1.json; tail -5 $O/bench_n1.err ;;
tetra)
  cd /tmp; rm -rf $O/tetra_kt $O/tetra_fetch $O/tetra_write
  T="python $R/bench.py --gpus 1 --workload tetra --steps 20 --warmup 5 --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tetra_kt -o kt -- $T > $O/tetra_kt.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tetra_fetch -o pmc -- $T > $O/tetra_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tetra_write -o pmc -- $T > $O/tetra_write.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/tetra_sq -o pmc -- $T > $O/tetra_sq.log 2>&1
  cd $R
  f=$(find $O/tetra_kt -name "*kernel_stats.csv" | head -1); cp $f $O/tetra_kernel_stats.csv; head -8 $f | cut -c1-200
  for d in tetra_fetch tetra_write tetra_sq; do python tools/summarize_pmc.py $O/$d $O/${d}_summary.csv 2>&1 | tail -6; done ;;
anib)    # C5 fragment mode at HEAD: the bench record, a kernel trace and one SQ pass of its steps
  timeout 900 python bench.py --gpus 1 --workload anib > $O/bench_anib.log 2> $O/bench_anib.err; echo "anib rc=$?"
  grep '^{' $O/bench_anib.log > $O/bench_anib_C5_n1.json; cut -c1-1500 $O/bench_anib_C5_n1.json
  cd /tmp; rm -rf $O/anib_kt $O/anib_sq
  A="python $R/bench.py --gpus 1 --workload anib --steps 1 --warmup 1 --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/anib_kt -o kt -- $A > $O/anib_kt.log 2>&1
  timeout -k 10 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $O/anib_sq -o pmc -- $A > $O/anib_sq.log 2>&1
  cd $R
  f=$(find $O/anib_kt -name "*kernel_stats.csv" | head -1); cp $f $O/anib_kernel_stats.csv; head -8 $f | cut -c1-200
  python tools/summarize_pmc.py $O/anib_sq $O/anib_sq_summary.csv 2>&1 | tail -8
  find $O/anib_kt $O/anib_sq -name "*.csv" -size +20M -delete ;;
anibpmc)   # HBM traffic of the C5 steps: FETCH_SIZE / WRITE_SIZE passes of the anib step's command (separate runs)
  cd /tmp; rm -rf $O/anib_fetch $O/anib_write
  A="python $R/bench.py --gpus 1 --workload anib --steps 1 --warmup 1 --no-cpu-baseline"
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/anib_fetch -o pmc -- $A > $O/anib_fetch.log 2>&1
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/anib_write -o pmc -- $A > $O/anib_write.log 2>&1
  cd $R
  for d in anib_fetch anib_write; do python tools/summarize_pmc.py $O/$d $O/${d}_summary.csv 2>&1 | tail -6 | cut -c1-200; done
  find $O/anib_fetch $O/anib_write -name "*.csv" -size +20M -delete ;;
final)   # what the driver runs at round end: smoke() and the default bench command (no flags), timed
  ( time timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -5 $O/smoke.log
  ( time timeout 1500 python bench.py > $O/bench_default.log 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "default bench rc=$?"; cat $O/bench_default.time
  grep '^{' $O/bench_default.log | cut -c1-600 ;;
summ)    # profiles/ on the box from what the steps before left (bench.py reads profiles/pmc_anim.json)
  python tools/summarize_round_profiles.py r06 | tail -30 ;;
cold)    # one cold end-to-end run of the whole C4 job from FASTA files on disk
  PYANI_BENCH_TMP=/tmp timeout -k 10 900 python bench.py --gpus 1 --cold-e2e > $O/cold_e2e.json 2> $O/cold_e2e.err; echo "cold rc=$?"
  cut -c1-1200 $O/cold_e2e.json; tail -3 $O/cold_e2e.err ;;
esac
done
du -sh $O
