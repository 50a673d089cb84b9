# usage (GPU box): bash tools/ab_env.sh VAR   -> the default bench line with VAR unset / VAR=1, alternating (value, certificate, host to host)
cd /root/repo
for rep in 1 2 3; do
for on in 0 1; do
  if [ $on = 1 ]; then export $1=1; else unset $1; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-traffic --cli-contigs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('$1=$on', 'step', d['ms_per_step'], 'sssp', s['sssp'], 'cert on top', d['certificate']['ms_on_top_of_run'], 'h2h', d['host_to_host']['ms_per_step'], 'two', d['two_batches_in_flight']['ms_per_step'], d['two_batches_in_flight']['host_to_host']['ms_per_step'], 'cfg5 h2h', d['strong_scaling_base']['host_to_host']['ms_per_step'], 'uncert', d['certificate']['contigs_not_certified_on_device'])"
done
done
