#!/bin/bash
# kernel timeline of a lone contig's steady-state runs (on the GPU box): tools/lone_trace.sh lambda|t4 -> gpurun_out/lone_trace_<name>.txt
name=${1:-t4}
export TMPDIR=/tmp
rm -rf /tmp/lt && rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o p -- python bench.py --workload $name --steps 6 --warmup 3 --no-extras --no-cpu --no-traffic --no-pipeline > /dev/null 2>&1
python tools/timeline.py $(find /tmp/lt -name p_kernel_trace.csv | head -1) 2 > gpurun_out/lone_trace_$name.txt 2>&1
