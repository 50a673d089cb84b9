cd $GRAFT_REPO_ROOT
o=gpurun_out/r06g; mkdir -p $o
for q in default 2 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for rep in 1 2; do
  python bench.py --no-extras --no-traffic --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['two_batches_in_flight']
print('queues $q', 'step', d['ms_per_step'], 'cert', d['certificate']['ms_per_step_with_run'], 'h2h', d['host_to_host']['ms_per_step'], 'two', t['ms_per_step'], 'two_h2h', t['host_to_host']['ms_per_step'])" | tee -a $o/queues.txt
  done
done
