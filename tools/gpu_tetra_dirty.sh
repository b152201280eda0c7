#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r02t; mkdir -p $O
timeout 600 python -m pytest tests/test_tetra_gpu.py tests/test_abi.py -q -x -m gpu --timeout 300 > $O/pytest_tetra.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_tetra.log
for d in 0 30 300 3000; do timeout 120 tools/microbench/count_bench 0.5 $d 2>&1 | grep "dirty\|BLOCK=512 PF=3" ; done | tee $O/count_dirty.txt
