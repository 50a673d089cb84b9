import sys, glob, gzip
sys.path.insert(0, '/root/repo')
import phanotate_amd as pa
from phanotate_amd import fasta
ann = pa.Annotator()
for f in sorted(glob.glob('/root/repo/tests/golden/*.fasta.gz')):
    recs = list(fasta.read_fasta(f))
    seqs = [s for _, s in recs]
    res = ann.annotate(seqs)
    for i, ((name, s), (st, g)) in enumerate(zip(recs, res)):
        gl = ann.globals(i)
        print(f.split('/')[-1], name[:20], len(s), 'status', st, 'genes', len(g), 'limbs', gl.n_limbs, 'kernel', gl.sssp_kernel, 'handed_back', gl.sssp_handed_back, 'sweeps', gl.sssp_sweeps)
