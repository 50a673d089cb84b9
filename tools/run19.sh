cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06_b > gpurun_out/collect_r06_b.log 2>&1
o=gpurun_out/prof_r06_b
python bench.py --workload lambda --steps 200 --warmup 20 > $o/bench_lambda.json 2>/dev/null
python bench.py --workload t4 --steps 200 --warmup 20 > $o/bench_t4.json 2>/dev/null
bash tools/lone_trace.sh lambda; bash tools/lone_trace.sh t4; cp gpurun_out/lone_trace_lambda.txt $o/timeline_lambda.txt; cp gpurun_out/lone_trace_t4.txt $o/timeline_t4.txt
python tools/h2h_parts.py > $o/h2h_parts.txt 2>&1
python tools/pipe_trace.py 1000 40 > $o/pipe_trace.txt 2>&1
python tools/h2h10k.py > $o/h2h10k.txt 2>&1
bash tools/batch_sizes.sh 8 64 256 512 1250 2000 > $o/batch_sizes.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $o/gputests.txt 2>&1; tail -3 $o/gputests.txt
tail -1 $o/bench.json | cut -c1-300; cat $o/batch_sizes.txt $o/pipe_trace.txt
