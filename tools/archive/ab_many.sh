# usage (GPU box): bash tools/ab_many.sh "<v1> <v2> ..." [contigs] [reps]  -> alternating runs of tmp_variants/libphx_<v>.so, all stages
cd /root/repo
cp phanotate_amd/libphx.so /tmp/d.so
n=${2:-1000}
for rep in $(seq 1 ${3:-3}); do
for v in $1; do
  cp tmp_variants/libphx_$v.so phanotate_amd/libphx.so
  timeout 300 python bench.py --contigs $n --steps 30 --warmup 3 --no-extras --no-cpu --no-traffic --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$v n=$n', d['ms_per_step'], ' '.join('%s %.4f' % (k, v) for k, v in s.items()))"
done
done
cp /tmp/d.so phanotate_amd/libphx.so
