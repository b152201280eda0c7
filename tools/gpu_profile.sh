#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command + PMC passes (separate runs, as the guide prescribes)
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/kt -o kt -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof/kt_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof/pmc_sq -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_sq.log 2>&1
cd $R
find gpurun_out/prof -name "*.csv" | head -30
for f in $(find gpurun_out/prof/kt -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
grep '^{' gpurun_out/prof/kt_bench.log | cut -c1-300
