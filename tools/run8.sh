cd $GRAFT_REPO_ROOT
o=gpurun_out/r06h; mkdir -p $o
cp phanotate_amd/libphx.so /tmp/new.so
timeout 600 bash tools/ab_libs.sh $o/ab.txt 3 phanotate_amd/libphx_base.so /tmp/new.so -- --steps 20 --warmup 3
timeout 300 bash tools/ab_libs.sh $o/ab1250.txt 2 phanotate_amd/libphx_base.so /tmp/new.so -- --steps 10 --warmup 3 --contigs 1250
timeout 900 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
