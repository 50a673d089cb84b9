for c in 256 1000; do PHX_DEBUG_SSSP=1 python bench.py --steps 1 --warmup 0 --no-cpu --contigs $c 2>&1 | grep "sssp contig" | tail -3; done
