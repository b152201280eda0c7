#!/bin/bash
# cluster stage split (ranges of clusters per wave): tests first (short timeouts), then A/B of the C4 bench
R=$(pwd); O=$R/gpurun_out/r02i; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_anim_gpu.py -x -q --timeout 300 > $O/pytest_anim.log 2>&1; rc=$?; tail -5 $O/pytest_anim.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-tetra"
for v in default "PYANI_ANIM_SPLIT_MIN=0" "PYANI_ANIM_WORKERS=1" "PYANI_ANIM_WORKERS=1 PYANI_ANIM_SPLIT_MIN=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  if [ "$v" = default ]; then timeout 600 $B > $O/$tag.json 2> $O/$tag.err; else timeout 600 env $v $B > $O/$tag.json 2> $O/$tag.err; fi
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], round(j['value']), {k:round(v) for k,v in j['roofline']['stage_ms'].items()}, j['config']['results_sha1_full_grid'])
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done
