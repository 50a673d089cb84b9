"""Development (run under torch.distributed.run, all ranks on GPU 0): are the genes of a context's FIRST run (buffers sized between kernels, fresh
device memory) those of its second run?  Several processes on one GPU hand each other's freed memory around, so a kernel that reads what
nobody wrote shows here and not on a box of its own."""
import os, sys
root = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import torch
import phanotate_amd as pa
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
seeds = list(range(rank, 10000, world))
seqs = [pa.synth_contig(i, 50000) for i in seeds]
ann = pa.Annotator(device=0, flags=tuple(sys.argv[2:])) if len(sys.argv) > 2 else pa.Annotator(device=0)
ann.upload(seqs); ann.run()
g1 = ann._download_flat() if hasattr(ann, "_download_flat") else ann.download_flat()
def summary(i):
    g = ann.globals(i)
    return dict(orf=g.n_orf, node=g.n_node, edge=g.n_edge, kern=g.sssp_kernel, back=g.sssp_handed_back, sweeps=g.sssp_sweeps, iters=g.sssp_iters, limbs=g.n_limbs, status=g.status, tie=g.tie)
first = [summary(i) for i in range(0, len(seqs), 50)]
ann.run()
g2 = ann._download_flat() if hasattr(ann, "_download_flat") else ann.download_flat()
second = [summary(i) for i in range(0, len(seqs), 50)]
same = g1[2].tobytes() == g2[2].tobytes() and g1[1].tobytes() == g2[1].tobytes()
if not same:
    k = 0
    for a_, b_ in zip(first, second):
        if a_ != b_ and k < 4:
            print("rank", rank, "contig", 50 * first.index(a_), "first", a_, "second", b_, flush=True); k += 1
nd = 0
if not same:
    for i in range(len(seqs)):
        a, b = g1[2][g1[1][i]:g1[1][i + 1]], g2[2][g2[1][i]:g2[1][i + 1]]
        nd += a.tobytes() != b.tobytes()
print("rank", rank, "first run == second run:", same, "contigs that differ:", nd, flush=True)
if not same:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle
    shown = 0
    for i in range(len(seqs)):
        a, b = g1[2][g1[1][i]:g1[1][i + 1]], g2[2][g2[1][i]:g2[1][i + 1]]
        if a.tobytes() == b.tobytes():
            continue
        o = oracle.run(seqs[i])
        ok1 = len(a) == len(o["gene_left"]) and np.array_equal(a["left"], o["gene_left"]) and np.array_equal(a["right"], o["gene_right"])
        ok2 = len(b) == len(o["gene_left"]) and np.array_equal(b["left"], o["gene_left"]) and np.array_equal(b["right"], o["gene_right"])
        fields = [f for f in ("left", "right", "strand", "frame", "score") if len(a) != len(b) or not np.array_equal(a[f], b[f])]
        print("rank", rank, "contig", i, "genes", len(a), len(b), "first run == oracle:", ok1, "second run == oracle:", ok2, "fields that differ:", fields, "status", g1[0][i], g2[0][i], flush=True)
        shown += 1
        if shown >= 3:
            break
