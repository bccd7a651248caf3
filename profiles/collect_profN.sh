cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
N=${1:-128}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profN -- python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-roofline --host-audio-steps 0 --streams $N --steps 6 --warmup 2 --spinup 2 > $O/profN.log 2>&1
S=$(find $O/profN -name "*kernel_stats.csv" | head -1); T=$(find $O/profN -name "*kernel_trace.csv" | head -1)
cp $S $O/bench_kernel_stats_prof$N.csv
python3 profiles/trace_gaps.py $T > $O/trace_busy_prof$N.txt
rm -rf $O/profN
sed -n 14,45p $O/trace_busy_prof$N.txt
