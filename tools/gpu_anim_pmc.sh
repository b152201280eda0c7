#!/bin/bash
# PMC pass (own run, no tracing flags besides the counters) over the ANIm probe: VALU issue utilisation of the DP kernels
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_anim
timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_anim -o pmc -- python $R/tools/anim_scale.py --n 25 --length 5000000 > $R/gpurun_out/pmc_anim.log 2>&1
ls $R/gpurun_out/pmc_anim | head
