#!/bin/bash
# refresh every measured artefact: TETRA profile passes, full bench (with CPU baseline), smoke, ANIm fixture report + scale probes
mkdir -p gpurun_out
bash tools/gpu_profile.sh > gpurun_out/rr_profile.log 2>&1
timeout 900 python bench.py > gpurun_out/rr_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/rr_bench.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/rr_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/rr_smoke.log
timeout 600 python tools/anim_fixture_report.py gpurun_out/rr_anim_fixture_report.json > gpurun_out/rr_anim_fixture.log 2>&1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/rr_kt_anim
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rr_kt_anim -o kt -- python $R/tools/anim_scale.py --n 50 --length 5000000 --out $R/gpurun_out/rr_anim_scale_50x5M.json > $R/gpurun_out/rr_anim_50.log 2>&1
cd $R
timeout 180 python tools/anim_scale.py --n 25 --length 5000000 --out gpurun_out/rr_anim_scale_25x5M.json > gpurun_out/rr_anim_25.log 2>&1
timeout 180 python tools/anim_scale.py --n 100 --length 5000000 --out gpurun_out/rr_anim_scale_100x5M.json > gpurun_out/rr_anim_100.log 2>&1
tail -2 gpurun_out/rr_bench.log | cut -c1-600; tail -2 gpurun_out/rr_smoke.log; tail -3 gpurun_out/rr_anim_fixture.log; for f in gpurun_out/rr_anim_*.log; do grep "^{" $f | cut -c1-160; done
