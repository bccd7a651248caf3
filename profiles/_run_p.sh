cd $GRAFT_REPO_ROOT; O=gpurun_out/r02b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fixture_replay.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -x -q > $O/tests_enc16.log 2>&1; tail -3 $O/tests_enc16.log
timeout 300 python profiles/phase_time_probe.py 2>/dev/null
for n in 1 2 4; do timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams $n --steps 32 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$n streams', j['ms_per_step'], j['value'])"; done
