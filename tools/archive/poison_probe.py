import sys, numpy as np
sys.path.insert(0, "/root/repo")
import phanotate_amd as pa
for n in (300, 1250):
    seqs = [pa.synth_contig(i, 50000) for i in range(n)]
    a = pa.Annotator(flags=("poison",))
    a.upload(seqs); a.run()
    g1 = a._download_flat()
    a.run()
    g2 = a._download_flat()
    nd = sum(g1[2][g1[1][i]:g1[1][i+1]].tobytes() != g2[2][g2[1][i]:g2[1][i+1]].tobytes() for i in range(n))
    print(n, "contigs; differ between the first and the second run:", nd, "statuses", np.unique(g1[0], return_counts=True))
    a.close()
