cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests_audio.log 2>&1; tail -3 $O/tests_audio.log
timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 64 --steps 16 --warmup 4 > $O/bench64.log 2>&1; tail -1 $O/bench64.log | cut -c1-400
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 64 --steps 16 --warmup 4 > $O/prof64.log 2>&1
T=$(find $O/prof64 -name "*kernel_trace.csv" | head -1)
python3 profiles/trace_gaps.py $T > $O/trace_busy_prof64.txt; rm -rf $O/prof64
head -20 $O/trace_busy_prof64.txt
