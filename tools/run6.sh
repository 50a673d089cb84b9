set -x
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06f; mkdir -p $o
export TMPDIR=/tmp
cp phanotate_amd/libphx.so /tmp/new.so
timeout 300 bash tools/ab_libs.sh $o/ab1250.txt 1 phanotate_amd/libphx_base.so /tmp/new.so -- --steps 10 --warmup 3 --contigs 1250
timeout 300 python tools/pipe_trace.py 1000 40 > $o/pipe_trace.txt 2>&1; cat $o/pipe_trace.txt
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/tl -o p -- python tools/pipe_trace.py 1000 12 > /dev/null 2>&1
find $o/tl -name "*.csv" | head
python tools/pipe_timeline.py $(find $o/tl -name "*kernel_trace.csv") $(find $o/tl -name "*memory_copy_trace.csv") > $o/pipe_timeline.txt 2>&1
rm -rf $o/tl
wc -l $o/pipe_timeline.txt
