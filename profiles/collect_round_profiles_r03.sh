# Round 3 profile collection (ONE gpurun call): default bench, rocprofv3 kernel traces of the one-stream and the 64-stream bench, the PMC passes over the dominant
# GEMV (roofline.traffic) and over the dense prefill GEMM (fabric traffic, matrix-pipe busy), beam 4 / 5 tokens / 30-minute stream / streams sweep.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python3 bench.py > $O/bench_v2.json 2> $O/bench_v2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --host-audio-steps 0 > $O/prof1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-roofline --host-audio-steps 0 --streams 64 --steps 16 --warmup 4 --spinup 4 > $O/prof64.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 profiles/roofline_probe.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 profiles/roofline_probe.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/dpmc_fetch -- python3 profiles/dense_pmc_probe.py > $O/dpmc_fetch.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/dpmc_mfma -- python3 profiles/dense_pmc_probe.py > $O/dpmc_mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -- python3 profiles/dense_pmc_probe.py > $O/dstats.log 2>&1
for d in prof1 prof64; do
  S=$(find $O/$d -name "*kernel_stats.csv" | head -1); T=$(find $O/$d -name "*kernel_trace.csv" | head -1)
  cp $S $O/bench_kernel_stats_${d}.csv
  python3 profiles/trace_gaps.py $T > $O/trace_busy_${d}.txt
done
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 profiles/roofline_traffic_reduce.py $F $W $O/roofline_traffic.json "round 3, $(date -u +%Y-%m-%d)"
DF=$(find $O/dpmc_fetch -name "*counter_collection.csv" | head -1); DM=$(find $O/dpmc_mfma -name "*counter_collection.csv" | head -1); DS=$(find $O/dstats -name "*kernel_stats.csv" | head -1)
python3 profiles/dense_pmc_reduce.py $DF $DM $DS $O/dense_pmc.json
python3 - <<PY
import csv
for f, k, o in (("$F", "gemm_skinny_kernel<1, 2, 5", "gemv_pmc_fetch_size.csv"), ("$W", "gemm_skinny_kernel<1, 2, 5", "gemv_pmc_write_size.csv"),
                ("$DF", "gemm_dense_kernel<5>", "dense_pmc_fetch_size.csv"), ("$DM", "gemm_dense_kernel<5>", "dense_pmc_mfma.csv")):
    rows = [r for r in csv.DictReader(open(f)) if k in r["Kernel_Name"]]
    if rows:
        with open("$O/" + o, "w", newline="") as fo:
            w = csv.DictWriter(fo, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
        print(o, len(rows))
PY
rm -rf $O/prof1 $O/prof64 $O/pmc_fetch $O/pmc_write $O/dpmc_fetch $O/dpmc_mfma $O/dstats
B="python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-roofline --host-audio-steps 0"
: > $O/streams_sweep.txt
for n in 1 2 4 8 16 32 64; do
  timeout 400 $B --streams $n --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print(f\"$n streams: {j['ms_per_step']} ms per chunk, {j['value']} xRT, p50 {j['p50_chunk_latency_ms']} p95 {j['p95_chunk_latency_ms']} host {j['host_ms_per_step']} ms kv {c.get('llm_kv_entries')} evictions {c.get('evictions_per_stream')}\")" >> $O/streams_sweep.txt
done
cat $O/streams_sweep.txt
timeout 400 $B --beam 4 > $O/bench_beam4.log 2>&1; tail -1 $O/bench_beam4.log | cut -c1-260
timeout 400 $B --gen-tokens 5 > $O/bench_g5.log 2>&1; tail -1 $O/bench_g5.log | cut -c1-260
timeout 600 $B --steps 1875 --warmup 8 > $O/bench_30min_stream.log 2>&1; tail -1 $O/bench_30min_stream.log | cut -c1-400
ls -la $O
