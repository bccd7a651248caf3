# Round 5 profile collection (ONE gpurun call): default bench, rocprofv3 kernel traces of the one-stream, beam-4, 64-stream and 64 x beam-4 bench,
# the PMC passes over the dominant GEMV (roofline.traffic), stream sweeps, the 30-minute stream.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/${TAG:-collect}   # TAG=final: the re-collection on the last build of the round
mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
B="python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-streams64-beam4 --no-multipliers --host-audio-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- $B > $O/prof1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb4 -- $B --no-roofline --beam 4 --steps 16 --warmup 4 --spinup 4 > $O/profb4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- $B --no-roofline --streams 64 --steps 12 --warmup 4 --spinup 4 > $O/prof64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64x4 -- $B --no-roofline --streams 64 --beam 4 --steps 6 --warmup 2 --spinup 2 > $O/prof64x4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 profiles/roofline_probe.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 profiles/roofline_probe.py > $O/pmc_write.log 2>&1
for d in prof1 profb4 prof64 prof64x4; do
  S=$(find $O/$d -name "*kernel_stats.csv" | head -1); T=$(find $O/$d -name "*kernel_trace.csv" | head -1)
  cp $S $O/bench_kernel_stats_${d}.csv
  python3 profiles/trace_gaps.py $T > $O/trace_busy_${d}.txt
done
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 profiles/roofline_traffic_reduce.py $F $W $O/roofline_traffic.json "round 5, $(date -u +%Y-%m-%d)"
python3 - <<PY
import csv
for f, k, o in (("$F", "gemm_skinny_kernel<1, 2, 5", "gemv_pmc_fetch_size.csv"), ("$W", "gemm_skinny_kernel<1, 2, 5", "gemv_pmc_write_size.csv")):
    rows = [r for r in csv.DictReader(open(f)) if k in r["Kernel_Name"]]
    if rows:
        with open("$O/" + o, "w", newline="") as fo:
            w = csv.DictWriter(fo, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
        print(o, len(rows))
PY
rm -rf $O/prof1 $O/profb4 $O/prof64 $O/prof64x4 $O/pmc_fetch $O/pmc_write
: > $O/streams_sweep.txt
for n in 1 2 4 8 16 32 48 64 96 128; do  # (greedy)
  timeout 600 $B --no-roofline --streams $n --steps 12 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print(f\"$n streams: {j['ms_per_step']} ms per chunk, {j['value']} xRT, p50 {j['p50_chunk_latency_ms']} p95 {j['p95_chunk_latency_ms']} host {j['host_ms_per_step']} ms kv {c.get('llm_kv_entries')} evictions {c.get('evictions_per_stream')}\")" >> $O/streams_sweep.txt
done
cat $O/streams_sweep.txt
: > $O/streams_sweep_beam4.txt
for n in 1 4 8 16 32 64 128; do
  timeout 600 $B --no-roofline --streams $n --beam 4 --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(f\"$n streams x beam 4: {j['ms_per_step']} ms per chunk, {j['value']} xRT, p50 {j['p50_chunk_latency_ms']} p95 {j['p95_chunk_latency_ms']} host {j['host_ms_per_step']} ms\")" >> $O/streams_sweep_beam4.txt
done
cat $O/streams_sweep_beam4.txt
timeout 600 $B --no-roofline --steps 1875 --warmup 8 > $O/bench_30min_stream.log 2>&1; tail -1 $O/bench_30min_stream.log | cut -c1-400
timeout 600 $B --no-roofline --streams 64 --beam 4 --steps 200 --warmup 4 > $O/soak_64x4_beams_200_steps.log 2>&1; tail -1 $O/soak_64x4_beams_200_steps.log | cut -c1-400
ls -la $O
