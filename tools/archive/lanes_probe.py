#!/usr/bin/env python3
"""What would splitting a batch over k contexts (own stream + host thread each) buy?  Development probe."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phanotate_amd as pa

def main():
    C_, L_ = 1000, 50000
    seqs = [pa.synth_contig(i, L_) for i in range(C_)]
    for k in (1, 2, 3, 4):
        anns = [pa.Annotator() for _ in range(k)]
        parts = [seqs[i::k] for i in range(k)]
        for a, p in zip(anns, parts): a.annotate(p)
        def work(a, n):
            for _ in range(n): a.run()
        for rep in range(2):
            ths = [threading.Thread(target=work, args=(a, 5)) for a in anns]
            t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            dt = (time.perf_counter() - t0) / 5
        print("lanes %d: %.3f ms per step (%.0f Mbp/s)" % (k, dt * 1e3, C_ * L_ / dt / 1e6))
        for a in anns: a.close()
main()
