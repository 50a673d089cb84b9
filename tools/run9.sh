cd $GRAFT_REPO_ROOT
o=gpurun_out/r06i; mkdir -p $o
export TMPDIR=/tmp
cp phanotate_amd/libphx.so /tmp/new.so
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
for v in new abl1 abl2 abl4 abl7; do
  if [ $v = new ]; then cp /tmp/new.so phanotate_amd/libphx.so; else cp tmp_variants/libphx_$v.so phanotate_amd/libphx.so; fi
  rm -rf /tmp/ks /tmp/sq
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python bench.py --steps 4 --warmup 1 --no-extras --no-pipeline --no-traffic > /dev/null 2>&1
  echo "== $v" >> $o/abl.txt
  grep "k_edges\|k_wave_plan<2, 0\|k_sssp_duo" $(find /tmp/ks -name "*kernel_stats.csv") | cut -d, -f1-4 >> $o/abl.txt
  timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/sq -o p -- python bench.py --steps 1 --warmup 0 --no-extras --no-pipeline --no-traffic > /dev/null 2>&1
  python tools/pmc_summary.py $(find /tmp/sq -name "*counter_collection.csv") 2>/dev/null | grep "k_edges" >> $o/abl.txt
done
cp /tmp/new.so phanotate_amd/libphx.so
cat $o/abl.txt
