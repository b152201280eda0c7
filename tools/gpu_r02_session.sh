#!/bin/bash
# Round-2 GPU session: GPU tests, the driver's bench command, rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE passes
# of the SAME bench command (shorter), everything under gpurun_out/r02/.  Usage: tools/gpu_r02_session.sh [tests|bench|prof]...
R=$(pwd); O=$R/gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
what=${@:-tests bench prof}
for w in $what; do
case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -15 $O/pytest_gpu.log ;;
bench)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.log 2> $O/bench_n1.err; echo "bench rc=$?"
  grep '^{' $O/bench_n1.log > $O/bench_n1.json; cut -c1-1500 $O/bench_n1.json; tail -5 $O/bench_n1.err ;;
prof)
  cd /tmp
  B="python $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-tetra"
  rm -rf $O/kt $O/pmc_fetch $O/pmc_write
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/kt_bench.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- $B > $O/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- $B > $O/pmc_write.log 2>&1
  cd $R
  python tools/summarize_anim_profiles.py r02 2>&1 | tail -40
  # the raw per-dispatch counter tables are large: keep only the summaries
  find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete ;;
esac
done
du -sh $O
