#!/bin/bash
# kernel trace of a lone contig's steady-state runs with segments: which of planner / segment solver / merge takes the solver stage
# usage: tools/seg_prof.sh lambda|t4  (on the GPU box; writes gpurun_out/seg_prof_<name>.txt)
name=${1:-lambda}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/segprof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/segprof -o p -- python $GRAFT_REPO_ROOT/tools/seg_time.py $name > /dev/null 2>&1
python - "$name" <<'PY'
import csv, glob, sys, os
f = glob.glob("/tmp/segprof/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/seg_prof_%s.txt" % sys.argv[1], "w")
for r in rows[:40]:
    out.write("%-90s calls %6s avg %10.1f ns\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])))
PY
