cd $GRAFT_REPO_ROOT; O=gpurun_out/r02b; mkdir -p $O
ISST_QKV_SLICES=2 timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -x -q > $O/tests_qs.log 2>&1; tail -3 $O/tests_qs.log
for n in 64 16; do for v in 1 2 4; do
ISST_QKV_SLICES=$v timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams $n --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('qkv_slices=$v $n streams', j['ms_per_step'], j['value'])"; done; done
for v in 1 2; do ISST_QKV_SLICES=$v timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --steps 24 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('qkv_slices=$v 1 stream', j['ms_per_step'], j['value'])"; done
