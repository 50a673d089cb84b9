cd $GRAFT_REPO_ROOT
o=gpurun_out/r06r; mkdir -p $o
for r in 1 2 3; do
for e in 1 0; do
  PHX_NO_ORF_ROWS=$e python bench.py --no-extras --no-traffic --no-pipeline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_orf_rows=$e', d['ms_per_step'], d['certificate']['ms_per_step_with_run'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" | tee -a $o/ab_orf_rows.txt
done; done
for e in 1 0; do PHX_NO_ORF_ROWS=$e python bench.py --no-extras --no-traffic --no-pipeline --steps 10 --warmup 3 --contigs 1250 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1250 no_orf_rows=$e', d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" | tee -a $o/ab_orf_rows.txt; done
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
