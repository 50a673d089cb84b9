#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/collect_profiles.sh <tag>
# Produces gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_fetch.csv, pmc_write.csv, pmc_sq.csv, bench.json, traffic.json}
# which are then copied into profiles/ (tracked).  Counter passes are separate runs, kernel-trace only.
set -u
tag=${1:-run}
out=gpurun_out/prof_$tag
mkdir -p $out/ks $out/f $out/w $out/sq
export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1"
BFULL="python bench.py --steps 20 --warmup 3"
$BFULL > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o p -- $B --no-extras --no-pipeline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -o p -- python bench.py --steps 1 --warmup 0 --no-extras --no-pipeline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -o p -- python bench.py --steps 1 --warmup 0 --no-extras --no-pipeline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq -o p -- python bench.py --steps 1 --warmup 0 --no-extras --no-pipeline > /dev/null 2>&1
cp $out/ks/p_kernel_stats.csv $out/kernel_stats.csv
mkdir -p $out/tl
rocprofv3 --kernel-trace --output-format csv -d $out/tl -o p -- python bench.py --steps 6 --warmup 1 --no-extras --no-cpu --no-traffic --no-pipeline-host > /dev/null 2>&1
python tools/timeline.py $out/ks/p_kernel_trace.csv 3 > $out/timeline.txt 2>&1
python tools/timeline2.py $out/tl/p_kernel_trace.csv > $out/timeline_two_batches.txt 2>&1
python tools/pmc_summary.py --json $out/traffic.json $out/f/p_counter_collection.csv $out/w/p_counter_collection.csv $out/sq/p_counter_collection.csv > $out/pmc_summary.txt
rm -rf $out/ks $out/f $out/w $out/sq $out/tl
ls -la $out
