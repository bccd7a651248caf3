# rocprofv3 kernel trace of the 64-stream step (BASELINE.json configs[2]) -> profiles/rNN/bench_kernel_stats_prof64*.csv, trace_busy_prof64*.txt
# usage (on the GPU box, through gpurun): bash profiles/collect_prof64.sh r03 [tag]
set -x
ROUND=${1:-r03}; TAG=${2:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64$TAG -- python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-roofline --host-audio-steps 0 --streams 64 --steps 16 --warmup 4 --spinup 4 > $O/prof64$TAG.log 2>&1
S=$(find $O/prof64$TAG -name "*kernel_stats.csv" | head -1); T=$(find $O/prof64$TAG -name "*kernel_trace.csv" | head -1)
cp $S $O/bench_kernel_stats_prof64$TAG.csv
python3 profiles/trace_gaps.py $T > $O/trace_busy_prof64$TAG.txt
rm -rf $O/prof64$TAG
cat $O/trace_busy_prof64$TAG.txt
