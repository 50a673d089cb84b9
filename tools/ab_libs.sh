#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/ab_libs.sh <out> <rounds> libA.so libB.so ... [-- bench args]
# Stage tables of several builds of libphx.so on ONE box (the boxes of the pool differ by a few per cent), taking turns.
out=$1; rounds=$2; shift 2
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" = "--" ] && shift
cp phanotate_amd/libphx.so /tmp/keep.so
mkdir -p $(dirname $out)
: > $out
for r in $(seq 1 $rounds); do
  for l in "${libs[@]}"; do
    cp $l phanotate_amd/libphx.so
    python bench.py --no-extras --no-traffic --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'], d['certificate']['ms_per_step_with_run'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" >> $out
  done
done
cp /tmp/keep.so phanotate_amd/libphx.so
cat $out
