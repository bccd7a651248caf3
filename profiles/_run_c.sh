cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_beam.py tests/test_gpu_fixture_replay.py -m gpu -x -q > $O/tests_pd2.log 2>&1; tail -3 $O/tests_pd2.log
for i in 1 2; do timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 64 --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('64 streams', j['ms_per_step'], j['value'])"; done
timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 16 --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('16 streams', j['ms_per_step'], j['value'])"
