cd $GRAFT_REPO_ROOT
o=gpurun_out/r06q; mkdir -p $o
cp phanotate_amd/libphx.so /tmp/new.so
timeout 600 bash tools/ab_libs.sh $o/ab.txt 3 tmp_variants/libphx_ranks0.so /tmp/new.so -- --steps 20 --warmup 3
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
