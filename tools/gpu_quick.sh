#!/bin/bash
# quick iteration: variant harness + parity tests + short bench
mkdir -p gpurun_out
timeout 120 tools/microbench/count_bench > gpurun_out/q_cb.log 2>&1
timeout 240 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/q_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/q_pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/q_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/q_bench.log
cat gpurun_out/q_cb.log; tail -5 gpurun_out/q_pytest.log; tail -2 gpurun_out/q_bench.log
