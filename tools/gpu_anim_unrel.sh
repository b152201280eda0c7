#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for only in unrelated related; do
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/anim_kt_$only -- python $GRAFT_REPO_ROOT/tools/anim_scale.py --n 50 --length 5000000 --only $only > $GRAFT_REPO_ROOT/gpurun_out/a_$only.log 2>&1
grep "^{" $GRAFT_REPO_ROOT/gpurun_out/a_$only.log | cut -c1-200
done
