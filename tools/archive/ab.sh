#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of the library on the same box (the boxes of the pool differ by a few per cent):
#   phanotate_amd/libphx.so (new) against phanotate_amd/libphx_base.so (built from the previous state, see below); prints
#   ms/step and the solver stage for each, alternating.   bash tools/ab.sh [rounds] [bench args...]
# Make the base:  git stash; make -C phanotate_amd/csrc; cp phanotate_amd/libphx.so phanotate_amd/libphx_base.so; git stash pop; make -C phanotate_amd/csrc
r=${1:-3}; shift
cp phanotate_amd/libphx.so /tmp/new.so
for i in $(seq $r); do
  for v in base new; do
    if [ $v = base ]; then cp phanotate_amd/libphx_base.so phanotate_amd/libphx.so; else cp /tmp/new.so phanotate_amd/libphx.so; fi
    python bench.py --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
cp /tmp/new.so phanotate_amd/libphx.so
