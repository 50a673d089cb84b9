#!/usr/bin/env python3
"""Like fuzz_gpu.py but few, long contigs (100-600 kb): concatenations of fuzz pieces.  python tools/fuzz_big.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuzz_gpu
_piece = fuzz_gpu.make
def make_big(rng):
    n = int(rng.randint(4, 18))
    return "".join(_piece(rng).lower() for _ in range(n)).replace("n" * 50, "acgt" * 12)
if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    fuzz_gpu.make = make_big
    sys.argv = [sys.argv[0], str(n), str(seed)]
    sys.exit(fuzz_gpu.main())
