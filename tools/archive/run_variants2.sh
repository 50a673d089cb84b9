#!/bin/bash
# on the GPU box: bench every tmp_variants/libphx_<i>.so, with the two-batches-in-flight lines
cat tmp_variants/list.txt
cp phanotate_amd/libphx.so /tmp/libphx_default.so
for f in tmp_variants/libphx_*.so; do
  v=${f##*_}; v=${v%.so}
  cp $f phanotate_amd/libphx.so
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-traffic --no-extras 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; t=d.get('two_batches_in_flight') or {}
print('variant $v', d['ms_per_step'], 'sssp', s['sssp'], 'features', s['features'], 'two', t.get('ms_per_step'), 'two h2h', (t.get('host_to_host') or {}).get('ms_per_step'), 'h2h', d['host_to_host']['ms_per_step'], 'genes', d['config']['genes_called_total'], 'bad', d['config']['contigs_with_error_status'])"
done
cp /tmp/libphx_default.so phanotate_amd/libphx.so
