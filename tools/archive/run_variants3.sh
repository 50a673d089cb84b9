#!/bin/bash
# on the GPU box: every stage of every tmp_variants/libphx_<i>.so (20 steps)
cat tmp_variants/list.txt
cp phanotate_amd/libphx.so /tmp/libphx_default.so
for f in tmp_variants/libphx_*.so; do
  v=${f##*_}; v=${v%.so}
  cp $f phanotate_amd/libphx.so
  timeout 300 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu --no-traffic --no-pipeline 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('variant $v', d['ms_per_step'], ' '.join('%s %.4f' % (k, v) for k, v in s.items()), 'genes', d['config']['genes_called_total'])"
done
cp /tmp/libphx_default.so phanotate_amd/libphx.so
