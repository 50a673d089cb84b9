#!/bin/bash
# round 4, second helping: more fuzz seeds and 30 000 more contigs of the benchmark series through the certificate
for s in $(seq 441 ${1:-700}); do timeout 900 python tools/fuzz_gpu.py 300 $s 2>&1 | tail -1 | cut -c1-420; done
python tools/cert_count.py 10000 30000 2500 2>&1 | tail -14
