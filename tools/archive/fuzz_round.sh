for s in 31 32 33 34 35 36; do timeout 600 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1; done
for s in 5 6; do timeout 900 python tools/fuzz_big.py $s 2>&1 | tail -2; done
python tools/validate_batch.py 2>&1 | tail -2
