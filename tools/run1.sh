set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06a/gputests.txt 2>&1; tail -5 gpurun_out/r06a/gputests.txt
cp phanotate_amd/libphx.so /tmp/new.so
timeout 600 bash tools/ab_libs.sh gpurun_out/r06a/ab.txt 2 phanotate_amd/libphx_base.so /tmp/new.so -- --steps 20 --warmup 3
