# Round 6 profile collection (ONE gpurun call): default bench, rocprofv3 kernel traces of the one-stream, beam-4, 64-stream, 128-stream and 64 x beam-4 bench,
# the PMC passes over the dominant GEMV (roofline.traffic) and over the dense prefill GEMM (MFMA busy, fabric reads), the per-kernel roofline table of the round
# (profiles/kernel_rooflines.py over THIS round's kernel stats), stream sweeps, the 30-minute stream.     usage: TAG=final bash profiles/collect_round_profiles_r06.sh
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/${TAG:-collect}
mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
B="python3 bench.py --no-cpu-baseline --no-streams64 --no-beam4 --no-streams64-beam4 --no-multipliers --host-audio-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- $B > $O/prof1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb4 -- $B --no-roofline --beam 4 --steps 16 --warmup 4 --spinup 4 > $O/profb4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- $B --no-roofline --streams 64 --steps 12 --warmup 4 --spinup 4 > $O/prof64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof128 -- $B --no-roofline --streams 128 --steps 6 --warmup 2 --spinup 2 > $O/prof128.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64x4 -- $B --no-roofline --streams 64 --beam 4 --steps 6 --warmup 2 --spinup 2 > $O/prof64x4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 profiles/roofline_probe.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 profiles/roofline_probe.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/dpmc_fetch -- python3 profiles/dense_pmc_probe.py > $O/dpmc_fetch.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/dpmc_mfma -- python3 profiles/dense_pmc_probe.py > $O/dpmc_mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstats -- python3 profiles/dense_pmc_probe.py > $O/dstats.log 2>&1
for d in prof1 profb4 prof64 prof128 prof64x4; do
  S=$(find $O/$d -name "*kernel_stats.csv" | head -1); T=$(find $O/$d -name "*kernel_trace.csv" | head -1)
  cp $S $O/bench_kernel_stats_${d}.csv
  python3 profiles/trace_gaps.py $T > $O/trace_busy_${d}.txt
done
python3 profiles/kernel_rooflines.py $O > $O/roofline_by_kernel.txt 2>&1
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python3 profiles/roofline_traffic_reduce.py $F $W $O/roofline_traffic.json "round 6, $(date -u +%Y-%m-%d)"
DF=$(find $O/dpmc_fetch -name "*counter_collection.csv" | head -1); DM=$(find $O/dpmc_mfma -name "*counter_collection.csv" | head -1); DS=$(find $O/dstats -name "*kernel_stats.csv" | head -1)
python3 profiles/dense_pmc_reduce.py $DF $DM $DS $O/dense_pmc.json
python3 - <<PY
import csv
for f, k, o in (("$F", "gemm_skinny_kernel<1, 1, 8", "gemv_pmc_fetch_size.csv"), ("$W", "gemm_skinny_kernel<1, 1, 8", "gemv_pmc_write_size.csv"),
                ("$DF", "gemm_dense_kernel<5>", "dense_pmc_fetch_size.csv"), ("$DM", "gemm_dense_kernel<5>", "dense_pmc_mfma.csv")):
    rows = [r for r in csv.DictReader(open(f)) if k in r["Kernel_Name"]]
    if rows:
        with open("$O/" + o, "w", newline="") as fo:
            w = csv.DictWriter(fo, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
        print(o, len(rows))
PY
rm -rf $O/prof1 $O/profb4 $O/prof64 $O/prof128 $O/prof64x4 $O/pmc_fetch $O/pmc_write $O/dpmc_fetch $O/dpmc_mfma $O/dstats
: > $O/streams_sweep.txt
for n in 1 2 4 8 16 32 64 128; do  # (greedy)
  timeout 600 $B --no-roofline --streams $n --steps 12 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print(f\"$n streams: {j['ms_per_step']} ms per chunk, {j['value']} xRT, p50 {j['p50_chunk_latency_ms']} p95 {j['p95_chunk_latency_ms']} host {j['host_ms_per_step']} ms kv {c.get('llm_kv_entries')} evictions {c.get('evictions_per_stream')}\")" >> $O/streams_sweep.txt
done
cat $O/streams_sweep.txt
: > $O/streams_sweep_beam4.txt
for n in 1 4 8 16 32 64 128; do
  timeout 600 $B --no-roofline --streams $n --beam 4 --steps 8 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(f\"$n streams x beam 4: {j['ms_per_step']} ms per chunk, {j['value']} xRT, p50 {j['p50_chunk_latency_ms']} p95 {j['p95_chunk_latency_ms']} host {j['host_ms_per_step']} ms\")" >> $O/streams_sweep_beam4.txt
done
cat $O/streams_sweep_beam4.txt
timeout 600 $B --no-roofline --steps 1875 --warmup 8 > $O/bench_30min_stream.log 2>&1; tail -1 $O/bench_30min_stream.log | cut -c1-400
ls -la $O
