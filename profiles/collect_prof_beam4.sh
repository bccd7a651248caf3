cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb4 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-roofline --host-audio-steps 0 --beam 4 --steps 16 --warmup 4 --spinup 4 > $O/profb4.log 2>&1
S=$(find $O/profb4 -name "*kernel_stats.csv" | head -1); T=$(find $O/profb4 -name "*kernel_trace.csv" | head -1)
cp $S $O/bench_kernel_stats_beam4.csv
python3 profiles/trace_gaps.py $T > $O/trace_busy_beam4.txt
rm -rf $O/profb4
head -45 $O/trace_busy_beam4.txt
