import gzip, os, sys
sys.path.insert(0, "/root/repo")
import phanotate_amd as pa
with gzip.open("/root/repo/tests/golden/NC_001416.1.fasta.gz", "rt") as f:
    lam = "".join(f.read().split("\n")[1:]).encode()
a = pa.Annotator(flags=("no_graph",))
a.annotate_flat([lam])
os.environ["PHX_DEBUG_FRONT"] = "1"
for _ in range(4):
    a.run()
