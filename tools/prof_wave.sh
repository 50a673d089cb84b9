# needs a library built with: make -C phanotate_amd/csrc EXTRA="-DPHX_DEV_REPORT -DWV_PROFILE"
for c in 1 256 1000; do echo "contigs $c"; PHX_DEBUG_WAVE=1 python bench.py --steps 1 --warmup 0 --no-extras --contigs $c 2>&1 | grep "wave" | tail -3; done
