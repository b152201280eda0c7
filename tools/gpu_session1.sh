#!/bin/bash
# first GPU session: microbench + parity tests + smoke + short bench + rocprof stats
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/s1_smi.log
nproc >> gpurun_out/s1_smi.log; lscpu | grep "Model name" >> gpurun_out/s1_smi.log
timeout 120 tools/microbench/lds_bench > gpurun_out/s1_ldsbench.log 2>&1
timeout 240 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1_pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/s1_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s1_smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/s1_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/s1_bench.log
tail -3 gpurun_out/s1_pytest.log; tail -2 gpurun_out/s1_smoke.log; tail -2 gpurun_out/s1_bench.log; cat gpurun_out/s1_ldsbench.log
