import time, os, sys
sys.path.insert(0,'.')
import phanotate_amd as pa
from phanotate_amd.fasta import Fasta
fa='/tmp/in10k.fasta'
with open(fa,'wb') as f:
    for i in range(10000):
        s=pa.synth_contig(i,50000)
        f.write(b">contig%05d synthetic\n"%i)
        f.write(b"\n".join(s[k:k+70] for k in range(0,len(s),70))+b"\n")
for th in ("1","4","8","16"):
    os.environ["PHX_HOST_THREADS"]=th
    for rep in range(2):
        t0=time.perf_counter(); F=Fasta(fa); t1=time.perf_counter(); print("threads %s parse %.3f s, n=%d"%(th,t1-t0,len(F))); F.close()
t0=time.perf_counter(); d=open(fa,'rb').read(); print("plain read %.3f"%(time.perf_counter()-t0))
