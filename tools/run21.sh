cd $GRAFT_REPO_ROOT
o=gpurun_out/r06s; mkdir -p $o
for r in 1 2 3; do
for e in 1 0; do
  PHX_NO_SIDE_SCORE=$e python bench.py --no-extras --no-traffic --no-pipeline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_side_score=$e', d['ms_per_step'], d['certificate']['ms_per_step_with_run'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" | tee -a $o/ab_side_score.txt
done; done
for e in 1 0; do PHX_NO_SIDE_SCORE=$e python bench.py --no-extras --no-traffic --steps 10 --warmup 3 --contigs 1250 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1250 no_side_score=$e', d['ms_per_step'], d['two_batches_in_flight']['ms_per_step'])" | tee -a $o/ab_side_score.txt; done
for e in 1 0; do PHX_NO_SIDE_SCORE=$e python bench.py --no-extras --no-traffic --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['two_batches_in_flight']; print('1000 no_side_score=$e', d['ms_per_step'], 'two', t['ms_per_step'], t['with_certificate']['ms_per_step'], 'h2h', d['host_to_host']['ms_per_step'], t['host_to_host']['ms_per_step'])" | tee -a $o/ab_side_score.txt; done
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
