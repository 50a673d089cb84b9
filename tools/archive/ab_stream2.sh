cd /root/repo
cp phanotate_amd/libphx.so /tmp/d.so
for rep in 1 2; do
for v in nostream stream late; do
  cp tmp_variants/libphx_$v.so phanotate_amd/libphx.so
  for n in 8 64 128 256 512 1000; do
    timeout 300 python bench.py --contigs $n --steps 30 --warmup 3 --no-extras --no-cpu --no-traffic --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$v n=$n', d['ms_per_step'], 'sssp', s['sssp'], 'plan', s['wave_plan'], 'fill', s['edges_fill'], 'genes', d['config'].get('genes_called_total'))"
  done
done
done
cp /tmp/d.so phanotate_amd/libphx.so
