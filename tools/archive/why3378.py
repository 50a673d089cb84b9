"""Why does the device not certify a contig (default: seed 3378 of the benchmark series)?   python tools/why3378.py [seed] [L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, phanotate_amd as pa, certify_probe
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import decimal_replay as dump
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3378
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
seqs = [pa.synth_contig(seed, L)]
ann = pa.Annotator(flags=("no_exact",)); ann.upload(seqs); ann.run()
print("cert", ann.certified().tolist(), "tie", ann.globals(0).tie)
nd, ed, dist = ann.nodes(0), ann.edges(0), ann.dist(0)
path = [int(x) for x in ann.path(0)[0]]
why, ok = certify_probe.certify(nd, ed, dist, path, ties=ann.globals(0).tie != 0)
print("prototype:", ok, why)
print("flagged after k_refine: %d of %d edges, %d of them with eps > 0" % (int(ed["inexact"].sum()), len(ed), int(((ed["inexact"] != 0) & (ed["err"] != 0)).sum())))
wdec = dump.decimal_weights(ann, 0, seqs[0])[2]
print("bounds violated:", certify_probe.bounds_hold(ed, wdec)[:5])
if why.startswith("edge"):
    k = int(why.split()[1]); e = ed[k]
    print("edge", k, "src", e["src"], nd[e["src"]]["pos"], "dst", e["dst"], nd[e["dst"]]["pos"], "w", e["w"], "inexact", e["inexact"], "d1", e["d1"], "d2", e["d2"], "err", e["err"], "W*-W", int(wdec[k] * 1000) - certify_probe.device_int(float(e["w"])), "src on path", int(e["src"]) in path, "dst on path", int(e["dst"]) in path)
    # the tree path edges with bounds
    fl = [j for j in range(len(ed)) if ed[j]["inexact"] and int(ed[j]["dst"]) in path and int(ed[j]["src"]) in path]
    for j in fl[:20]:
        print("  flagged edge on the path", j, "w %.6g" % ed[j]["w"], "D", ed[j]["d1"] + ed[j]["d2"], "err", ed[j]["err"], "W*-W", int(wdec[j] * 1000) - certify_probe.device_int(float(ed[j]["w"])))
