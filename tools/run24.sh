cd $GRAFT_REPO_ROOT
o=gpurun_out/r06v; mkdir -p $o
for w in lambda t4; do python bench.py --workload $w --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'])" | tee -a $o/lone.txt; done
for c in 8 64 256 1000; do python bench.py --no-extras --no-traffic --no-pipeline --steps 20 --warmup 3 --contigs $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c contigs', d['ms_per_step'])" | tee -a $o/lone.txt; done
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
