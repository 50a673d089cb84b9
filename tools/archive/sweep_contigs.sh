for c in ${CONTIGS:-256 512 1000 2000 4000 8000}; do
python bench.py --steps 3 --warmup 1 --no-cpu --contigs $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print($c, d['ms_per_step'], d['value'], d['pcie_inclusive_Mbp_s'], round(s['sssp'],3), round(s['features'],3), round(s['edges_fill'],3))"
done
