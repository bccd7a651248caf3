cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 300 python profiles/mid_kw8_probe.py > $O/mid_kw8_probe.txt 2>&1; cat $O/mid_kw8_probe.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests_kw8.log 2>&1; tail -3 $O/tests_kw8.log
for n in 64 16 1; do timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams $n --steps 24 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$n streams', j['ms_per_step'], j['value'])"; done
