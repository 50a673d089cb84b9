# usage (GPU box): bash tools/ab_two.sh <variantA> <variantB> [contigs]   -> alternating runs of tmp_variants/libphx_<v>.so
cd /root/repo
cp phanotate_amd/libphx.so /tmp/d.so
n=${3:-1000}
for rep in 1 2 3; do
for v in $1 $2; do
  cp tmp_variants/libphx_$v.so phanotate_amd/libphx.so
  timeout 300 python bench.py --contigs $n --steps 30 --warmup 3 --no-extras --no-cpu --no-traffic --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$v n=$n', d['ms_per_step'], 'sssp', s['sssp'], 'plan', s['wave_plan'], 'fill', s['edges_fill'], 'kernel', d['roofline']['avg_launch_ms'])"
done
done
cp /tmp/d.so phanotate_amd/libphx.so
