#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI on a FASTA of N synthetic 50 kb contigs (north_star: 10 000 contigs annotated end to end):
    python tools/cli_e2e.py [contigs] [extra phanotate.py arguments...]      -> one JSON line: parse / GPU / format / write seconds as phanotate.py reports them
(PHX_CLI_TIMING), the wall-clock of the whole process, output size."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
extra = sys.argv[2:]
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, "in.fasta")
    t0 = time.perf_counter()
    with open(fa, "wb") as f:
        for i in range(n):
            s = pa.synth_contig(i, 50000)
            f.write(b">contig%05d synthetic\n" % i)
            f.write(b"\n".join(s[k:k + 70] for k in range(0, len(s), 70)) + b"\n")
    t_gen = time.perf_counter() - t0
    out = os.path.join(td, "out.tsv")
    res = []
    for rep in range(2):  # the second run finds the file in the page cache
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py"), "-o", out, fa] + extra, capture_output=True, text=True, env=dict(os.environ, PHX_CLI_TIMING="1"))
        wall = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-500:]
        t = [json.loads(l[len("PHX_CLI_TIMING "):]) for l in r.stderr.splitlines() if l.startswith("PHX_CLI_TIMING ")][-1]
        t.update(process_wall_s=round(wall, 3), contigs=n, fasta_bytes=os.path.getsize(fa), output_bytes=os.path.getsize(out), Mbp_s_process_wall=round(t["bases"] / wall / 1e6, 1))
        res.append(t)
    print(json.dumps({"fasta_written_s": round(t_gen, 2), "first_run": res[0], "second_run": res[1]}))
