cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06_a > gpurun_out/collect_r06_a.log 2>&1
o=gpurun_out/prof_r06_a
python bench.py --workload lambda --steps 200 --warmup 20 > $o/bench_lambda.json 2>/dev/null
python bench.py --workload t4 --steps 200 --warmup 20 > $o/bench_t4.json 2>/dev/null
python tools/h2h_parts.py > $o/h2h_parts.txt 2>&1
python tools/pipe_trace.py 1000 40 > $o/pipe_trace.txt 2>&1
tail -3 $o/bench.json | cut -c1-1500
cat $o/h2h_parts.txt $o/pipe_trace.txt
ls $o
