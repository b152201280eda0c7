#!/bin/bash
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_unrel
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_unrel -o kt -- python $R/tools/anim_scale.py --n 200 --length 5000000 --only unrelated > $R/gpurun_out/unrel.log 2>&1
grep "^{" $R/gpurun_out/unrel.log | cut -c1-200
