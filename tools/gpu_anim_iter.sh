#!/bin/bash
# Every command carries its own SHORT timeout: a hang inside one of them must not run into gpurun's limit (which is
# clamped to the GPU budget left and is charged in full).
# ANIm iteration: parity tests, then the related-pairs scale probes with a kernel trace of the larger one
mkdir -p gpurun_out
timeout 240 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a_pytest.log
timeout 180 python tools/anim_scale.py --n 25 --length 5000000 --out gpurun_out/anim_scale_25x5M.json > gpurun_out/a_25x5M.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/anim_kt_50 -- python $GRAFT_REPO_ROOT/tools/anim_scale.py --n 50 --length 5000000 --out $GRAFT_REPO_ROOT/gpurun_out/anim_scale_50x5M.json > $GRAFT_REPO_ROOT/gpurun_out/a_50x5M.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/a_pytest.log; tail -1 gpurun_out/a_25x5M.log; tail -1 gpurun_out/a_50x5M.log
f=$(ls gpurun_out/anim_kt_50/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" | cut -d, -f1-4
