"""A fuzz contig on which the library and the oracle disagree: who is right?   python tools/fuzz_why.py <seed> [n]
Decides with decimal.Decimal itself (dump.python_resolve: the reference's integers, python ints)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, phanotate_amd as pa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import decimal_replay as dump
from oracle import oracle
import fuzz_gpu
seed = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.RandomState(seed)
seqs = [fuzz_gpu.make(rng) for _ in range(n)]
ann = pa.Annotator()
for b0 in range(0, n, 100):
    part = seqs[b0:b0 + 100]
    res = ann.annotate(part)
    cert = ann.certified()
    for i, (st, genes) in enumerate(res):
        o = oracle.run(part[i])
        if o["status"] != (st if st < 0 else 0):
            gl = ann.globals(i)
            print("contig %d len %d: status library %d, oracle %d (oracle wide %s, path sum %s bits); limbs %d" % (b0 + i, len(part[i]), st, o["status"], o.get("wide"), abs(o["path_dist"]).bit_length() if o["status"] == 0 and len(o["path"]) else None, gl.n_limbs))
            continue
        if o["status"] < 0 or st < 0:
            continue
        mine = [(int(x["left"]), int(x["right"]), int(x["strand"])) for x in genes]
        his = list(zip(o["gene_left"].tolist(), o["gene_right"].tolist(), o["gene_strand"].tolist()))
        if mine != his:
            gl = ann.globals(i)
            py = [t[:3] for t in dump.python_resolve(ann, i, part[i] if isinstance(part[i], str) else part[i].decode())]
            raw = ann.download_flat(exact=False); rg = raw[2][raw[1][i]:raw[1][i + 1]]
            rawl = [(int(x["left"]), int(x["right"]), int(x["strand"])) for x in rg]
            print("contig %d len %d limbs %d kernel %d tie %d cert %d oracle wide %s: library == Decimal replay %s, oracle == Decimal replay %s, raw device == library %s; genes %d vs %d"
                  % (b0 + i, len(part[i]), gl.n_limbs, gl.sssp_kernel, gl.tie, int(cert[i]), o.get("wide"), mine == py, his == py, rawl == mine, len(mine), len(his)))
            d = [(a, b) for a, b in zip(mine, his) if a != b][:3]
            print("  first differences (library, oracle):", d)
            ann.annotate(part)  # (download_flat(exact=False) reset the state)
