#!/bin/bash
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_seed
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_seed -o pmc -- python $R/tools/anim_scale.py --n 100 --length 5000000 --only unrelated > $R/gpurun_out/pmc_seed.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_seed2 -o pmc -- python $R/tools/anim_scale.py --n 100 --length 5000000 --only unrelated > $R/gpurun_out/pmc_seed2.log 2>&1
ls $R/gpurun_out/pmc_seed $R/gpurun_out/pmc_seed2
