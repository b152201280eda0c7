#!/bin/bash
# end-of-round session: all GPU tests, smoke(), the driver's bench command, profiles, two small A/B checks
R=$(pwd); O=$R/gpurun_out/r02; mkdir -p $O
bash tools/gpu_r02_session.sh tests bench prof
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 120 tools/microbench/count_bench 0.5 3000 2>&1 | grep "BLOCK=512 PF=3" | tee $O/count_dirty3000.txt
AB_STEPS=2 bash tools/gpu_ab.sh r02n "PYANI_ANIM_WORKERS=3"
