# auxiliary bench runs of a round (one gpurun call): streams sweep, beam 4, 5 tokens per chunk, the 30-minute stream (BASELINE.json configs[4])
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline"
: > $O/streams_sweep_v3.txt
for n in 1 2 4 8 16 32 64; do
  timeout 400 $B --streams $n --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print(f\"$n streams: {j['ms_per_step']} ms per chunk, {j['value']} xRT, p50 {j['p50_chunk_latency_ms']} p95 {j['p95_chunk_latency_ms']} kv {c.get('llm_kv_entries')} evictions {c.get('evictions_per_stream')}\")" >> $O/streams_sweep_v3.txt
done
cat $O/streams_sweep_v3.txt
timeout 400 $B --beam 4 > $O/bench_v3_beam4.log 2>&1; tail -1 $O/bench_v3_beam4.log | cut -c1-260
timeout 400 $B --gen-tokens 5 > $O/bench_v3_g5.log 2>&1; tail -1 $O/bench_v3_g5.log | cut -c1-260
timeout 600 $B --steps 1875 --warmup 8 > $O/bench_v3_30min_stream.log 2>&1; tail -1 $O/bench_v3_30min_stream.log | cut -c1-400
