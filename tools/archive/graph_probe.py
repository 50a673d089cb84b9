"""ms per phx_run with and without HIP-graph replay, stage profiling off (development probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phanotate_amd as pa
seqs = [pa.synth_contig(i, 50000) for i in range(1000)]
direct = len(sys.argv) > 1 and sys.argv[1] == "direct"
ann = pa.Annotator(flags=("no_graph",) if direct else ())
ann.annotate(seqs)
for _ in range(3): ann.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): ann.run()
torch.cuda.synchronize()
print("direct" if direct else "graph", "%.4f ms per run" % ((time.perf_counter() - t0) / 20 * 1e3))
