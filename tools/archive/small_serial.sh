#!/bin/bash
# on the GPU box: stand-alone duration of every kernel for a batch of 20000 short contigs (300-3000 bp).   bash tools/small_serial.sh
export TMPDIR=/tmp; d=/tmp/kser_$$; mkdir -p $d
cat > $d/run.py <<'PY'
import sys
sys.path.insert(0, '.')
import numpy as np, phanotate_amd as pa
rng = np.random.RandomState(7)
seqs = [pa.synth_contig(i, int(rng.randint(300, 3000))) for i in range(20000)]
a = pa.Annotator(); a.annotate(seqs); a.run(); a.run()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $d -o p -- python $d/run.py > /dev/null 2>&1
python - "$d" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
t = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:44]
    t.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print("  ".join("%s %.0f" % (k.split("(")[0].replace("void ", ""), v[-1]) for k, v in t.items()))
PY
rm -rf $d
