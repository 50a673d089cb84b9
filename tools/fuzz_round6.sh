#!/bin/bash
# round 6, after the coded gap edges (ESRC_F_GAP: the duo feeder, k_inorder, k_seg_join, k_certify read the gap table; k_edges_expand completes the rows
# for the fallback kernels the fuzz contigs take: modes 1 and 3) and the certificate behind phx_run_async: against the ORACLE; every line must say "0 mismatches"
for s in 641 642 643 644; do timeout 900 python tools/fuzz_lone.py 300 $s 2>&1 | tail -1 | cut -c1-260; done
for s in 661 662 663; do timeout 900 python tools/fuzz_big.py 40 $s 2>&1 | tail -1 | cut -c1-220; done
for s in 631 632 633 634 635 636; do timeout 600 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-200; done
timeout 1200 python tools/fuzz_params.py 1000 671 2>&1 | tail -1 | cut -c1-250
timeout 900 python tools/seg_fuzz.py 300 40 6 2>&1 | tail -2 | cut -c1-260
timeout 600 python tools/validate_batch.py 1000 2>&1 | tail -1
