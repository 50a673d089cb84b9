#!/bin/bash
# on the GPU box: step time of resident batches beyond one contig per SIMD -> stdout, one line per size:
#   contigs  ms/step  Mbp/s  solver stage ms  |  two batches in flight: ms per batch  Mbp/s        bash tools/batch_sizes.sh [sizes...]
for c in ${@:-1250 2000 4000}; do
python bench.py --steps 5 --warmup 2 --no-extras --contigs $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('two_batches_in_flight',{})
print($c, d['ms_per_step'], d['value'], round(d['stage_ms_per_step']['sssp'],4), t.get('ms_per_step'), t.get('value'))"
done
