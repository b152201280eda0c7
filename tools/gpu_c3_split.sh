#!/bin/bash
# Development aid: C3 with the cluster stage forced to the split form (prep kernel with 16 waves per unit + wave kernel)
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt
PYANI_ANIM_SPLIT_CLUSTER=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/tools/anim_scale.py --n 200 --length 5000000 > /tmp/c3.log 2>&1
grep "^{" /tmp/c3.log | cut -c1-120; grep -o '"results_sha1": "[0-9a-f]*' /tmp/c3.log
python $R/tools/kstats.py "/tmp/kt/*kernel_stats.csv" 8
