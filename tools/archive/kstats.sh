#!/bin/bash
# on the GPU box: per-kernel average durations of the default bench (rocprofv3 --kernel-trace --stats) -> gpurun_out/kstats_<tag>.csv
tag=${1:-x}; export TMPDIR=/tmp; mkdir -p gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -o p -- python bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2>&1
cp gpurun_out/ks_$tag/p_kernel_stats.csv gpurun_out/kstats_$tag.csv; rm -rf gpurun_out/ks_$tag
cut -d, -f1-4 gpurun_out/kstats_$tag.csv | head -30
