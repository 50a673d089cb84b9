cd $GRAFT_REPO_ROOT
o=gpurun_out/r06m; mkdir -p $o
timeout 900 bash tools/ab_libs.sh $o/ab.txt 3 tmp_variants/libphx_head.so tmp_variants/libphx_hoist.so -- --steps 20 --warmup 3
cp tmp_variants/libphx_hoist.so phanotate_amd/libphx.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fused or batch" > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
