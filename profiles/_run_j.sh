cd $GRAFT_REPO_ROOT; O=gpurun_out/r02b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "launch_free" 2>&1 | tail -2
for n in 64 32 16; do for v in 0 1; do
ISST_FUSE_REDUCE=$v timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams $n --steps 16 --warmup 4 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('fuse_reduce=$v $n streams', j['ms_per_step'], j['value'])"; done; done
