"""Development: certify stage time against the number of contigs in the batch (latency of one workgroup's chain or throughput?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phanotate_amd as pa
for n in (64, 250, 500, 1000, 2000):
    seqs = [pa.synth_contig(i, 50000) for i in range(n)]
    ann = pa.Annotator()
    ann.annotate_flat(seqs)
    for _ in range(2): ann.run()
    ann.set_profiling(True); ann.stage_ms(reset=True)
    for _ in range(5): ann.run()
    st = ann.stage_ms(reset=True)
    print(n, {k: round(v[0] / 5, 4) for k, v in st.items() if k in ("certify", "inorder", "sssp", "features", "edges_fill")})
    ann.close()
