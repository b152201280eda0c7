#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r02k; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_anim_gpu.py tests/test_anim_c3_gpu.py tests/test_anim_oos_gpu.py -x -q --timeout 600 > $O/pytest_anim.log 2>&1; rc=$?; tail -5 $O/pytest_anim.log
[ $rc -ne 0 ] && exit 1
bash tools/gpu_ab.sh r02k default "PYANI_ANIM_NO_MIRROR=1" "PYANI_ANIM_WORKERS=1" "PYANI_ANIM_WORKERS=1 PYANI_ANIM_NO_MIRROR=1"
