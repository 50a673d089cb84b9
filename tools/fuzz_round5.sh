#!/bin/bash
# round 5 (k_sssp_duo): fuzz contigs, lone contigs (the feeder waits on the planner), drawn flags, long concatenations, the whole benchmark batch; every line must say "0 mismatches"
for s in $(seq 501 530); do timeout 600 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-200; done
for s in 21 22 23; do timeout 900 python tools/fuzz_lone.py 300 $s 2>&1 | tail -1 | cut -c1-250; done
for s in 51 52 53 54; do timeout 900 python tools/fuzz_params.py 1500 $s 2>&1 | tail -1 | cut -c1-250; done
for s in 41 42; do timeout 900 python tools/fuzz_big.py 40 $s 2>&1 | tail -1 | cut -c1-200; done
timeout 600 python tools/validate_batch.py 1000 2>&1 | tail -1
