cd $GRAFT_REPO_ROOT
o=gpurun_out/r06t; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
for s in 651 652; do timeout 900 python tools/fuzz_lone.py 300 $s 2>&1 | tail -1 | cut -c1-260; done | tee $o/fuzz.txt
timeout 600 python tools/fuzz_gpu.py 300 653 2>&1 | tail -1 | cut -c1-200 | tee -a $o/fuzz.txt
python bench.py --workload lambda --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lambda', d['ms_per_step'])"
python bench.py --workload t4 --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('t4', d['ms_per_step'])"
python bench.py --no-extras --no-traffic --no-pipeline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d['ms_per_step'], d['certificate']['ms_per_step_with_run'])"
