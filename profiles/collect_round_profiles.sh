set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python3 bench.py > $O/bench_v1.json 2> $O/bench_v1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --no-cpu-baseline --no-streams64 > $O/prof1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 64 --steps 16 --warmup 4 > $O/prof64.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 profiles/roofline_probe.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 profiles/roofline_probe.py > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -30
for d in prof1 prof64; do
  S=$(find $O/$d -name "*kernel_stats.csv" | head -1); T=$(find $O/$d -name "*kernel_trace.csv" | head -1)
  cp $S $O/bench_kernel_stats_${d}.csv
  python3 profiles/trace_gaps.py $T > $O/trace_busy_${d}.txt
  rm -f $T
done
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 profiles/roofline_traffic_reduce.py $F $W $O/roofline_traffic.json "round 2, $(date -u +%Y-%m-%d)"
python3 - <<PY
import csv
for f in ("$F", "$W"):
    rows = [r for r in csv.DictReader(open(f)) if "gemm_skinny_kernel<1, 2, 5" in r["Kernel_Name"]]
    out = f.replace("counter_collection.csv", "gemv_only.csv")
    with open(out, "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    print(out, len(rows))
PY
cp $(find $O/pmc_fetch -name "*gemv_only.csv") $O/gemv_pmc_fetch_size.csv; cp $(find $O/pmc_write -name "*gemv_only.csv") $O/gemv_pmc_write_size.csv
rm -rf $O/prof1 $O/prof64 $O/pmc_fetch $O/pmc_write
ls -la $O
