#!/bin/bash
# Round-3 GPU session steps (everything under gpurun_out/r03/).  Usage: tools/gpu_r03_session.sh step...
#   exact   the exactness tests of the postnuc extender (fixtures, concordance, synthetic statement)
#   tests   the whole -m gpu suite
#   c3 / c4 bench lines (C3: 200 genomes; C4: the driver's default, few steps)
#   bench   the driver's command
#   prof    rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the bench command (short)
R=$(pwd); O=$R/gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
export PYANI_DEV_KNOBS=1      # the library honours its development variables (PYANI_PN_STATS, PYANI_ANIM_WORKERS) only under this switch
for w in "$@"; do
case $w in
exact)
  timeout 1500 python -m pytest tests/test_anim_oos_gpu.py tests/test_zz_concordance_gpu.py tests/test_anim_gpu.py -m gpu -q --timeout 900 > $O/pytest_exact.log 2>&1
  echo "pytest rc=$?" >> $O/pytest_exact.log; tail -25 $O/pytest_exact.log ;;
tests)
  timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -25 $O/pytest_gpu.log ;;
stats)   # one C3 grid with the engines' counters (PYANI_PN_STATS) and one worker, so that the launch times are not overlapped
  PYANI_PN_STATS=1 PYANI_ANIM_WORKERS=1 timeout 900 python bench.py --gpus 1 --genomes 200 --seed 20250228 --rows-per-step 200 --steps 1 --warmup 0 --no-tetra --no-cpu-baseline > $O/bench_c3_stats.log 2> $O/bench_c3_stats.err; echo "stats rc=$?"
  grep '^{' $O/bench_c3_stats.log | cut -c1-1500; grep "pn-stats" $O/bench_c3_stats.err | tail -8 ;;
stats4)  # one C4 step (the driver's step: 100 rows) with the counters and the per-item clocks of the units / forced kernels
  PYANI_PN_STATS=1 PYANI_ANIM_WORKERS=1 timeout 900 python bench.py --gpus 1 --steps 1 --warmup 0 --no-tetra --no-cpu-baseline > $O/bench_c4_stats.log 2> $O/bench_c4_stats.err; echo "stats4 rc=$?"
  grep '^{' $O/bench_c4_stats.log | cut -c1-1500; grep "pn-stats" $O/bench_c4_stats.err | tail -16 ;;
anib)
  timeout 1200 python -m pytest tests/test_anib_gpu.py tests/test_zz_concordance_gpu.py -m gpu -q --timeout 900 > $O/pytest_anib.log 2>&1; echo "pytest rc=$?" >> $O/pytest_anib.log
  tail -30 $O/pytest_anib.log; cp gpurun_out/anib_blast_agreement.json $O/ 2>/dev/null ;;
steps)   # C4 at two step sizes: how much a smaller step (shorter launches) costs
  for R in 100 250; do
    timeout 900 python bench.py --gpus 1 --rows-per-step $R --steps 2 --warmup 1 --no-tetra --no-cpu-baseline > $O/bench_c4_R$R.log 2> $O/bench_c4_R$R.err; echo "R=$R rc=$?"
    python - <<PY
import json
d=json.loads([l for l in open("$O/bench_c4_R$R.log") if l.startswith("{")][-1])
print("R=$R", "pairs/s", round(d["value"]), "ms/step", round(d["ms_per_step"]), "grid s", round(d["config"]["wall_s_grid"],1), "cold first step s", d["config"]["cold_first_step_s"], d["roofline"]["stage_ms"])
PY
  done ;;
c3)
  timeout 900 python bench.py --gpus 1 --genomes 200 --seed 20250228 --steps 3 --warmup 1 --no-tetra > $O/bench_c3.log 2> $O/bench_c3.err; echo "c3 rc=$?"
  grep '^{' $O/bench_c3.log > $O/bench_c3.json; cut -c1-2500 $O/bench_c3.json; tail -5 $O/bench_c3.err ;;
c4)
  timeout 1200 python bench.py --gpus 1 --steps 4 --warmup 1 --no-tetra --no-cpu-baseline > $O/bench_c4.log 2> $O/bench_c4.err; echo "c4 rc=$?"
  grep '^{' $O/bench_c4.log > $O/bench_c4.json; cut -c1-2500 $O/bench_c4.json; tail -5 $O/bench_c4.err ;;
bench)
  timeout 1500 python bench.py --gpus 1 --steps ${BENCH_STEPS:-20} --warmup ${BENCH_WARMUP:-5} > $O/bench_n1.log 2> $O/bench_n1.err; echo "bench rc=$?"
  grep '^{' $O/bench_n1.log > $O/bench_n1.json; cut -c1-3000 $O/bench_n1.json; tail -5 $O/bench_n1.err ;;
c5)      # C5 fragment mode (word tier on)
  timeout 900 python bench.py --gpus 1 --workload anib > $O/bench_c5.log 2> $O/bench_c5.err; echo "c5 rc=$?"
  grep '^{' $O/bench_c5.log > $O/bench_c5.json; cut -c1-2500 $O/bench_c5.json; tail -3 $O/bench_c5.err ;;
workers) # C4 steps with 2 and 4 host workers (streams): how much of a launch's sequential tail the others' kernels cover
  for W in 2 4; do
    PYANI_ANIM_WORKERS=$W timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-tetra --no-cpu-baseline > $O/bench_c4_W$W.log 2> $O/bench_c4_W$W.err; echo "W=$W rc=$?"
    python - <<PY2
import json
d=json.loads([l for l in open("$O/bench_c4_W$W.log") if l.startswith("{")][-1])
print("W=$W", "pairs/s", round(d["value"]), "ms/step", round(d["ms_per_step"]), "cold first step s", d["config"]["cold_first_step_s"])
PY2
  done ;;
kt)      # kernel trace only, one worker (launches not overlapped): per-kernel durations of one C4 step
  cd /tmp
  rm -rf $O/kt1
  PYANI_ANIM_WORKERS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -o kt -- python $R/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-tetra > $O/kt1_bench.log 2>&1
  cd $R
  f=$(find $O/kt1 -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-200
  find $O/kt1 -name "*kernel_trace.csv" -size +20M -delete ;;
prof)
  cd /tmp
  B="python $R/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-tetra"
  rm -rf $O/kt $O/pmc_fetch $O/pmc_write
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/kt_bench.log 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- $B > $O/pmc_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- $B > $O/pmc_write.log 2>&1
  cd $R
  python tools/summarize_anim_profiles.py r03 2>&1 | tail -40
  find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete ;;
esac
done
du -sh $O
