#!/bin/bash
# round 4, last build: 150 seeds x 300 fuzz contigs, 20 seeds x 1500 contigs under drawn flags, 300 lone contigs x 3 runs — against the oracle
for s in $(seq 2001 2150); do timeout 900 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-420; done
for s in $(seq 21 40); do timeout 900 python tools/fuzz_params.py 25 $s 2>&1 | tail -1; done
timeout 900 python tools/fuzz_lone.py 300 11 2>&1 | tail -1
