#!/bin/bash
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_c4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_c4 -o kt -- python $R/tools/anim_scale.py --n 1000 --length 5000000 --seed 20250301 > $R/gpurun_out/c4.log 2>&1
grep "^{" $R/gpurun_out/c4.log | cut -c1-200
rm -f $R/gpurun_out/kt_c4/kt_kernel_trace.csv
