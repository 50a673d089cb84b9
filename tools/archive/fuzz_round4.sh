#!/bin/bash
# round 4: fuzz contigs against the oracle with the exactness machinery on (k_refine + k_certify + the host re-solve inside the library)
for s in $(seq 401 ${1:-440}); do timeout 900 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-420; done
for s in 41 42; do timeout 900 python tools/fuzz_big.py 40 $s 2>&1 | tail -1 | cut -c1-300; done
