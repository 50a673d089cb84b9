"""End-to-end (H2D + kernels + D2H) timing of repeated annotate() calls, with and without torch's HIP runtime (development probe)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0)
import phanotate_amd as pa
for C_ in (1000, 2000):
    seqs = [pa.synth_contig(i, 50000) for i in range(C_)]
    ann = pa.Annotator()
    for rep in range(5):
        t0 = time.perf_counter(); ann.upload(seqs); t1 = time.perf_counter(); ann.run(); t2 = time.perf_counter(); r = ann.download(); t3 = time.perf_counter()
        print(C_, "call %d: upload %.1f ms run %.1f ms download %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    ann.close()
