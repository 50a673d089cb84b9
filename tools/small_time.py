import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, phanotate_amd as pa
for n,lo,hi in ((20000,300,3000),(100000,1000,3000),(5000,5000,20000)):
    rng=np.random.RandomState(7)
    seqs=[pa.synth_contig(i, int(rng.randint(lo,hi))) for i in range(n)]
    bp=sum(len(s) for s in seqs)
    a=pa.Annotator(); a.annotate(seqs); a.run(); a.set_profiling(True); a.stage_ms(reset=True)
    t=time.perf_counter(); 
    for _ in range(3): a.run()
    dt=(time.perf_counter()-t)/3
    st=a.stage_ms()
    print(n, "contigs", bp/1e6, "Mbp: %.2f ms/run = %.0f Mbp/s"%(dt*1e3, bp/dt/1e6), {k:round(v[0]/3,3) for k,v in st.items() if v[1]})
    a.close()
