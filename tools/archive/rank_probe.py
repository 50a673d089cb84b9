"""Development: what a rank of the sharded bench sees (run under torch.distributed.run): uncertified contigs and edge-tap failures of its shard."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import phanotate_amd as pa
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "pipe"
seeds = list(range(rank, 10000, world))
seqs = [pa.synth_contig(i, 50000) for i in seeds]
if mode == "pipe":
    pipe = pa.Pipeline(device=0, depth=2); ann = pipe.anns[0]
else:
    ann = pa.Annotator(device=0)
ann.upload(seqs); ann.run()
cert = ann.certified()
unc = np.nonzero(cert == 0)[0]
bad = 0
for i in list(unc[:5]) + list(range(0, len(seqs), 97)):
    try:
        ann.edges(int(i))
    except Exception as e:
        bad += 1
        if bad <= 2: print("rank", rank, "contig", i, str(e)[-160:])
print("rank", rank, "contigs", len(seqs), "uncertified", [seeds[i] for i in unc][:10], len(unc), "tap failures", bad, flush=True)
g1 = ann.download_flat(exact=False)
ann.run()
cert2 = ann.certified()
g2 = ann.download_flat(exact=False)
bad2 = 0
for i in list(unc[:5]) + list(range(0, len(seqs), 97)):
    try:
        ann.edges(int(i))
    except Exception as e:
        bad2 += 1
print("rank", rank, "second run: uncertified", int((cert2 == 0).sum()), "tap failures", bad2, "genes equal to the first run:", g1[2].tobytes() == g2[2].tobytes(), flush=True)
