set -x
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06d; mkdir -p $o
export TMPDIR=/tmp
cp phanotate_amd/libphx.so /tmp/new.so
timeout 600 bash tools/ab_libs.sh $o/ab.txt 2 phanotate_amd/libphx_base.so /tmp/new.so -- --steps 20 --warmup 3
timeout 300 bash tools/ab_libs.sh $o/ab1250.txt 1 phanotate_amd/libphx_base.so /tmp/new.so -- --steps 10 --warmup 3 --contigs 1250
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/ks -o p -- python bench.py --steps 5 --warmup 2 --no-extras --no-pipeline --no-traffic > /dev/null 2>&1
cp $o/ks/*/p_kernel_stats.csv $o/kernel_stats.csv 2>/dev/null || cp $o/ks/p_kernel_stats.csv $o/kernel_stats.csv
rm -rf $o/ks
head -40 $o/kernel_stats.csv
timeout 300 python tools/h2h_parts.py > $o/h2h_parts.txt 2>&1; cat $o/h2h_parts.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "certif or golden or refine or neartie" > $o/gputests_subset.txt 2>&1; tail -3 $o/gputests_subset.txt
