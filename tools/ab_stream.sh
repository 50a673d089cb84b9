cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_case or negative or neartie" 2>&1 | tail -3
cp phanotate_amd/libphx.so /tmp/d.so
for v in nostream stream nostream stream; do
  cp tmp_variants/libphx_$v.so phanotate_amd/libphx.so
  for w in lambda t4; do
    timeout 300 python bench.py --workload $w --steps 50 --warmup 5 --no-extras --no-cpu --no-traffic --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$v $w', d['ms_per_step'], 'sssp', s['sssp'], 'plan', s['wave_plan'], 'fill', s['edges_fill'])"
  done
done
cp /tmp/d.so phanotate_amd/libphx.so
