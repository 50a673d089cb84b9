#!/bin/bash
# round 4, after the solver learnt to follow its planner (batches of up to 800 contigs): the fuzz generator's contigs in batches of 100
# (streamed), lone contigs (streamed, the solver really waits on the counter), the big-contig fuzz, 20 000 short contigs in one batch
# (not streamed) — all against the oracle
for s in $(seq 1001 1040); do timeout 900 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-420; done
for s in 51 52; do timeout 900 python tools/fuzz_big.py 40 $s 2>&1 | tail -1 | cut -c1-300; done
timeout 900 python tools/many_small.py 20000 2>&1 | tail -1
timeout 900 python tools/fuzz_lone.py 150 7 2>&1 | tail -1
