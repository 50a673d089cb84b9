for c in 256 512 768 900 1000 1024 1280 1536 2048; do
python bench.py --steps 3 --warmup 1 --no-cpu --contigs $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print($c, d['ms_per_step'], {k:round(v,3) for k,v in s.items()})"
done
