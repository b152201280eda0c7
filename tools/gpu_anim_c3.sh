#!/bin/bash
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_c3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_c3 -o kt -- python $R/tools/anim_scale.py --n 200 --length 5000000 --out $R/gpurun_out/rr_anim_scale_200x5M.json > $R/gpurun_out/c3.log 2>&1
grep "^{" $R/gpurun_out/c3.log | cut -c1-200
