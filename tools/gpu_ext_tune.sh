#!/bin/bash
# Development aid: kernel-time totals of the 50 x 5 Mb probe for a few settings of the lane kernel's tail rule.
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for cfg in "24 64" "32 64" "32 256" "40 1024" "16 32" "48 128"; do
  set -- $cfg
  rm -rf /tmp/kt
  PYANI_EXT_TAIL_LANES=$1 PYANI_EXT_TAIL_BLOCKS=$2 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/anim_scale.py --n 50 --length 5000000 > /tmp/o.log 2>&1
  echo "== lanes<$1 blocks>=$2: $(grep -o '"results_sha1": "[0-9a-f]*' /tmp/o.log | cut -c18-30)"
  python $GRAFT_REPO_ROOT/tools/kstats.py "/tmp/kt/*/*kernel_stats.csv" 40 | grep "extdp_lane\|anim_extend\|total"
done
