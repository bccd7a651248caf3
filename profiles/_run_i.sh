cd $GRAFT_REPO_ROOT; O=gpurun_out/r02b; mkdir -p $O
ISST_FUSE_REDUCE=1 timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_beam.py -m gpu -x -q > $O/tests_fr.log 2>&1; tail -4 $O/tests_fr.log
for rep in 1 2; do for v in 0 1; do
ISST_FUSE_REDUCE=$v timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 64 --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('fuse_reduce=$v 64 streams', j['ms_per_step'], j['value'])"; done; done
for rep in 1 2; do for v in 0 1; do
ISST_FUSE_REDUCE=$v timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --steps 32 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('fuse_reduce=$v 1 stream', j['ms_per_step'], j['value'])"; done; done
for v in 0 1; do ISST_FUSE_REDUCE=$v timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams 16 --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('fuse_reduce=$v 16 streams', j['ms_per_step'], j['value'])"; done
