cd $GRAFT_REPO_ROOT; O=gpurun_out/r02b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests_ks.log 2>&1; tail -3 $O/tests_ks.log
for n in 64 1; do timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams $n --steps 24 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$n streams', j['ms_per_step'], j['value'])"; done
