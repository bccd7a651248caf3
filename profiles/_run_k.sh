cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 64 --steps 16 --warmup 4 > $O/prof64.log 2>&1
T=$(find $O/prof64 -name "*kernel_trace.csv" | head -1)
python3 profiles/trace_gaps.py $T > $O/trace_busy_prof64.txt; rm -rf $O/prof64
sed -n 13,50p $O/trace_busy_prof64.txt
