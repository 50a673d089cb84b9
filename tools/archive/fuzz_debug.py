import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuzz_gpu
from oracle import oracle
import phanotate_amd as pa
seed = int(sys.argv[1]); idxs = [int(x) for x in sys.argv[2:]]
rng = np.random.RandomState(seed)
seqs = [fuzz_gpu.make(rng) for _ in range(300)]
ann = pa.Annotator()
for i in idxs:
    s = seqs[i]
    (status, genes), = ann.annotate([s])
    g = ann.globals(0)
    o = oracle.run(s)
    p, dist = ann.path(0)
    print("contig", i, "len", len(s), "kernel", g.sssp_kernel, "limbs", g.n_limbs, "handed_back", g.sssp_handed_back, "gpu dist", dist, "oracle dist", o["path_dist"], "equal", dist == o["path_dist"])
    gl = [(int(a), int(b), int(c)) for a, b, c in zip(genes["left"], genes["right"], genes["strand"])]
    ol = [(int(a), int(b), int(c)) for a, b, c in zip(o["gene_left"], o["gene_right"], o["gene_strand"])]
    d = [(x, y) for x, y in zip(gl, ol) if x != y]
    print("  differing genes (gpu, oracle):", d[:6], "n", len(d))
    # weights of the differing genes
    sc = {(int(a), int(b)): float(w) for a, b, w in zip(genes["left"], genes["right"], genes["score"])}
    so = {(int(a), int(b)): float(w) for a, b, w in zip(o["gene_left"], o["gene_right"], o["gene_score"])}
    for x, y in d[:3]: print("   gpu gene", x, sc.get(x[:2]), " oracle gene", y, so.get(y[:2]))
