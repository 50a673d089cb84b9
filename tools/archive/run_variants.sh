#!/bin/bash
# on the GPU box: bench every tmp_variants/libphx_<i>.so
cat tmp_variants/list.txt
cp phanotate_amd/libphx.so /tmp/libphx_default.so
for f in tmp_variants/libphx_*.so; do
  v=${f##*_}; v=${v%.so}
  cp $f phanotate_amd/libphx.so
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extras 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('variant $v', d['value'], d['ms_per_step'], 'sssp', s['sssp'], 'features', s['features'], 'efill', s['edges_fill'], 'ecount', s['edges_count'], 'certify', s.get('certify'), 'inorder', s.get('inorder'), 'genes', d['config']['genes_called_total'], 'bad', d['config']['contigs_with_error_status'])"
done
cp /tmp/libphx_default.so phanotate_amd/libphx.so
