cd $GRAFT_REPO_ROOT
o=gpurun_out/r06k; mkdir -p $o
cp phanotate_amd/libphx.so /tmp/new.so
timeout 600 bash tools/ab_libs.sh $o/ab.txt 3 phanotate_amd/libphx_base.so /tmp/new.so -- --steps 20 --warmup 3
timeout 200 python tools/round_probe.py 800 > $o/round_probe.txt 2>&1; cat $o/round_probe.txt
PHX_NO_EAGER_CERT=1 timeout 200 python tools/round_probe.py 800 > $o/round_probe_noeager.txt 2>&1; cat $o/round_probe_noeager.txt
cp tmp_variants/libphx_duoprof.so phanotate_amd/libphx.so
timeout 200 python tools/duo_balance.py > $o/duo_balance.txt 2>&1; cat $o/duo_balance.txt
cp /tmp/new.so phanotate_amd/libphx.so
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
