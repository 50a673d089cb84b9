"""Development: one contig with a very long ORF (argv: codons), resident: limbs, kernel, ms per run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, phanotate_amd as pa
rng = np.random.RandomState(42)
sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
quiet = [c for c in sense if c not in ("atg", "gtg", "ttg")]
for arg in sys.argv[1:] or ["2200", "3000", "5500", "9000", "q9000"]:  # q: in-frame starts at 1 % instead of 5 % of the codons
    ncod = int(arg.lstrip("q"))
    body = "".join(rng.choice(sense, ncod)) if not arg.startswith("q") else "".join("atg" if rng.rand() < 0.01 else quiet[rng.randint(len(quiet))] for _ in range(ncod))
    seq = (pa.synth_contig(900, 20000).decode() + "atg" + body + "taa" + pa.synth_contig(1900, 20000).decode()).encode()
    a = pa.Annotator(flags=("no_certify",)); a.annotate([seq])
    for _ in range(3): a.run()
    a.set_profiling(True); a.stage_ms(reset=True)
    for _ in range(5): a.run()
    st = a.stage_ms(); g = a.globals(0)
    print("%s codons: limbs %d kernel %d status %d: solver stage %.3f ms, step %.3f ms" % (arg, g.n_limbs, g.sssp_kernel, g.status, st["sssp"][0] / 5, sum(v[0] for v in st.values()) / 5))
    a.close()
