set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python3 bench.py > $O/bench_v5.json 2> $O/bench_v5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 > $O/prof1.log 2>&1
S=$(find $O/prof1 -name "*kernel_stats.csv" | head -1); T=$(find $O/prof1 -name "*kernel_trace.csv" | head -1)
cp $S $O/bench_kernel_stats_prof1.csv; python3 profiles/trace_gaps.py $T > $O/trace_busy_prof1.txt; rm -rf $O/prof1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-roofline --host-audio-steps 0 --streams 64 --steps 16 --warmup 4 --spinup 4 > $O/prof64.log 2>&1
S=$(find $O/prof64 -name "*kernel_stats.csv" | head -1); T=$(find $O/prof64 -name "*kernel_trace.csv" | head -1)
cp $S $O/bench_kernel_stats_prof64.csv; python3 profiles/trace_gaps.py $T > $O/trace_busy_prof64.txt; rm -rf $O/prof64
bash profiles/collect_prof_beam4.sh > /dev/null 2>&1
head -30 $O/trace_busy_prof64.txt
tail -c 400 $O/bench_v5.json
