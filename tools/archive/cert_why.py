"""Development: which benchmark contigs k_certify leaves uncertified, and what the python statement of the certificate says about them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import phanotate_amd as pa
import certify_probe
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
from decimal_replay import decimal_weights
from decimal_check import solve
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ann = pa.Annotator()
ann.upload(seqs); ann.run()
cert = ann.certified()
bad = np.nonzero(cert == 0)[0]
print("uncertified:", bad.tolist())
for i in bad[:8]:
    nd, ed, dist = ann.nodes(int(i)), ann.edges(int(i)), ann.dist(int(i))
    path = [int(x) for x in ann.path(int(i))[0]]
    why, ok = certify_probe.certify(nd, ed, dist, path)
    k = int(why.split()[1].rstrip(":")) if why.startswith("edge") else -1
    extra = ""
    if k >= 0:
        e = ed[k]
        extra = " src %d (pos %d) -> dst %d (pos %d) w %.6g; on path: %s %s" % (e["src"], nd[e["src"]]["pos"], e["dst"], nd[e["dst"]]["pos"], e["w"], int(e["src"]) in path, int(e["dst"]) in path)
    nd2, ed2, wdec = decimal_weights(ann, int(i), seqs[int(i)].decode())
    same = solve(nd2, ed2, wdec) == path
    print(i, "prototype:", ok, why, extra, "| Decimal path same:", same, "| tie", ann.globals(int(i)).tie)
