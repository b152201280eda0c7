#!/bin/bash
# Development aid: build the library with -DPGA_DP_STATS locally first (tools/build_dpstats.sh), then run this on the GPU box.
python tools/anim_scale.py --n ${1:-25} --length 5000000 2>&1 | grep -v "^{" | cut -c1-600
