cd $GRAFT_REPO_ROOT
o=gpurun_out/r06o; mkdir -p $o
cp phanotate_amd/libphx.so /tmp/new.so
cp tmp_variants/libphx_hwid.so phanotate_amd/libphx.so
timeout 200 python tools/duo_hwid.py > $o/duo_hwid.txt 2>&1; cat $o/duo_hwid.txt
cp /tmp/new.so phanotate_amd/libphx.so
