#!/bin/bash
# A/B runs of the C4 bench under environment variants: tools/gpu_ab.sh OUTDIR "VAR=1 VAR2=x" "..." ...
R=$(pwd); O=$R/gpurun_out/$1; shift; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps ${AB_STEPS:-2} --warmup 1 --no-cpu-baseline --no-tetra"
for v in "$@"; do
  tag=$(echo "$v" | tr ' =' '__')
  if [ "$v" = default ]; then timeout 600 $B > $O/$tag.json 2> $O/$tag.err; else timeout 600 env $v $B > $O/$tag.json 2> $O/$tag.err; fi
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], round(j['value']), {k.replace('anim_','').replace('_kernels','').replace('_kernel',''):round(v) for k,v in j['roofline']['stage_ms'].items()}, j['config']['results_sha1_full_grid'][:8])
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done
