for c in 1 256 1000; do echo "contigs $c"; PHX_DEBUG_WAVE=1 python bench.py --steps 1 --warmup 0 --no-cpu --contigs $c 2>&1 | grep "wave" | tail -3; done
