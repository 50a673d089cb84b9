for s in $(seq 101 120); do timeout 600 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-120; done
for s in 11 12 13; do timeout 900 python tools/fuzz_big.py $s 2>&1 | tail -1 | cut -c1-160; done
