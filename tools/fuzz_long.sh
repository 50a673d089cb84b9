#!/bin/bash
# >= 20 000 fuzz contigs + long concatenations against the oracle; every line must say "0 mismatches"
for s in $(seq 301 370); do timeout 600 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-200; done
for s in 31 32 33 34; do timeout 900 python tools/fuzz_big.py 40 $s 2>&1 | tail -1 | cut -c1-200; done
