#!/bin/bash
# on the GPU box: stand-alone duration of every kernel of one step (a --pmc pass serialises the kernels) -> stdout.   bash tools/kserial.sh
export TMPDIR=/tmp; d=/tmp/kser_$$; mkdir -p $d
rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $d -o p -- python bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2>&1
python - "$d" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
t = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:44]
    t.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print("  ".join("%s %.0f" % (k.split("(")[0].replace("void ", ""), sum(v) / len(v)) for k, v in t.items()))
PY
rm -rf $d
