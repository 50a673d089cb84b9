#!/bin/bash
# SQ instruction counters per kernel (run on the GPU box)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/sq -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-extras ${1:-} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sq2 -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-extras ${1:-} > /dev/null 2>&1
python /root/repo/tools/pmc_summary.py /tmp/sq/p_counter_collection.csv /tmp/sq2/p_counter_collection.csv 2>/dev/null | grep -i "${2:-sssp}"
