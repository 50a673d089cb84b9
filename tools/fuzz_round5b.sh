#!/bin/bash
# round 5, after the segments (phx_sssp_seg.inc): lone contigs and few long concatenations run through them, against the ORACLE; batches of 300 and the
# benchmark batch take the one-sweep kernels; every line must say "0 mismatches"
for s in 31 32 33 34 35 36; do timeout 900 python tools/fuzz_lone.py 300 $s 2>&1 | tail -1 | cut -c1-260; done
for s in 61 62 63 64; do timeout 900 python tools/fuzz_big.py 40 $s 2>&1 | tail -1 | cut -c1-220; done
for s in 531 532 533 534 535 536; do timeout 600 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-200; done
for s in 71 72; do timeout 900 python tools/fuzz_params.py 1500 $s 2>&1 | tail -1 | cut -c1-250; done
timeout 600 python tools/validate_batch.py 1000 2>&1 | tail -1
