"""Development (library built with EXTRA=-DDUO_HWID): which SIMD the hardware gave the solver and the feeder wavefront of every contig of the benchmark
batch: how many solver wavefronts share a SIMD (two solvers on one SIMD take turns at its VALU; a solver beside a feeder has it nearly to itself)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ann = pa.Annotator(flags=("no_certify",))
ann.annotate_flat(seqs)
ann.run(); ann.run(); ann._download_flat()
g = [ann.globals(i) for i in range(n)]
def place(hw, xcc):
    return (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)  # XCC, SE, SH, CU, SIMD
sol = [place(x.rbs_training_count[20], x.rbs_training_count[22]) for x in g]
fee = [place(x.rbs_training_count[21], x.rbs_training_count[23]) for x in g]
cs, cf = collections.Counter(sol), collections.Counter(fee)
simds = set(cs) | set(cf)
print("SIMDs in use: %d; solvers per SIMD: %s; feeders per SIMD: %s" % (len(simds), dict(collections.Counter(cs[s] for s in simds)), dict(collections.Counter(cf[s] for s in simds))))
print("CUs in use: %d; XCCs: %s" % (len({s[:4] for s in simds}), dict(collections.Counter(s[0] for s in sol))))
same = sum(1 for a, b in zip(sol, fee) if a == b)
print("contigs whose two wavefronts share a SIMD: %d; share a CU: %d" % (same, sum(1 for a, b in zip(sol, fee) if a[:4] == b[:4])))
t = np.array([x.sssp_iters for x in g])
print("first contigs:", sol[:6], fee[:6])
# with -DDUO_PROFILE as well: time per phase of the solvers that share their SIMD with another solver against the others
tt = np.array([x.rbs_background_count[6] for x in g]) / 100.0
if tt.sum() > 0:
    shared = np.array([cs[s] >= 2 for s in sol])
    per = tt / np.maximum(t, 1)
    print("solver us per phase: alone on its SIMD (beside a feeder) %.4f (n = %d), sharing it with another solver %.4f (n = %d)" % (per[~shared].mean(), (~shared).sum(), per[shared].mean(), shared.sum()))
    print("solver us: alone mean %.1f max %.1f; sharing mean %.1f max %.1f; slowest ten share: %s" % (tt[~shared].mean(), tt[~shared].max(), tt[shared].mean(), tt[shared].max(), shared[np.argsort(-tt)[:10]].tolist()))
