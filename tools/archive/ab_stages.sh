#!/bin/bash
# Run ON THE GPU BOX: like tools/ab.sh, but prints the stage times of the two builds side by side.   bash tools/ab_stages.sh
cp phanotate_amd/libphx.so /tmp/new.so
for v in base new; do
  if [ $v = base ]; then cp phanotate_amd/libphx_base.so phanotate_amd/libphx.so; else cp /tmp/new.so phanotate_amd/libphx.so; fi
  python bench.py --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
cp /tmp/new.so phanotate_amd/libphx.so
