"""The C replay of the reference's Decimal chain (csrc/phx_dec.c + phx_exact.inc) against the same replay with Python's own decimal
(tests/decimal_replay.py), edge by edge and contig by contig, on contigs no fixture holds:
  * phx_dump_text == dump.dump_lines: every edge's str(Decimal weight * 1000), byte for byte;
  * with the certificate's bounds inflated (cert_tight) every contig goes through the host re-solve: its genes == dump.python_resolve.
    python tools/exact_crosscheck.py [n_benchmark] [n_fuzz] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import phanotate_amd as pa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import decimal_replay as dump
from fuzz_gpu import make
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
rng = np.random.RandomState(seed)
seqs = [pa.synth_contig(20000 + i, 50000).decode() for i in range(nb)]
while len(seqs) < nb + nf:
    s = make(rng).lower()
    if 400 <= len(s) <= 20000 and set(s) <= set("acgt"):
        seqs.append(s)
t0 = time.time()
ann = pa.Annotator(flags=("cert_tight",))
n_edges = n_dump_bad = n_res = n_res_bad = n_contigs = 0
for b0 in range(0, len(seqs), 40):
    part = seqs[b0:b0 + 40]
    st, offs, genes = ann.annotate_flat(part)
    cert = ann.certified()
    for i in range(len(part)):
        if st[i] < 0 or ann.globals(i).n_node <= 2:
            continue
        n_contigs += 1
        text = ann.dump_text(i)
        lines = dump.dump_lines(ann, i, part[i])
        n_edges += len(lines)
        if ("".join(l + "\n" for l in lines)).encode() != text:
            n_dump_bad += 1
            print("DUMP DIFFERS contig %d" % (b0 + i))
        if cert[i] == 2:
            n_res += 1
            py = dump.python_resolve(ann, i, part[i])
            g = genes[offs[i]:offs[i + 1]]
            if [(int(x["left"]), int(x["right"]), int(x["strand"]), int(x["frame"]), float(x["score"])) for x in g] != py:
                n_res_bad += 1
                print("RE-SOLVE DIFFERS contig %d" % (b0 + i))
print("exact crosscheck: %d contigs (%d benchmark-series, %d fuzz), %d edges: dump text differs on %d contigs; %d contigs solved again on the host (bounds inflated), %d differ from the Python replay; %.0f s"
      % (n_contigs, nb, nf, n_edges, n_dump_bad, n_res, n_res_bad, time.time() - t0))
