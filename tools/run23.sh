cd $GRAFT_REPO_ROOT
o=gpurun_out/r06u; mkdir -p $o
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
for w in lambda t4; do
PHX_NO_ORF_ROWS=$1 PHX_NO_SIDE_SCORE=$2 python bench.py --workload $w --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w no_orf_rows=$1 no_side_score=$2', d['ms_per_step'])" | tee -a $o/lone.txt
done
for c in 8 64 256; do
PHX_NO_ORF_ROWS=$1 PHX_NO_SIDE_SCORE=$2 python bench.py --no-extras --no-traffic --no-pipeline --steps 20 --warmup 3 --contigs $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c contigs no_orf_rows=$1 no_side_score=$2', d['ms_per_step'])" | tee -a $o/lone.txt
done; done
