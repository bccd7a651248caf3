cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q > $O/tests_devaudio.log 2>&1; tail -3 $O/tests_devaudio.log
for f in "" "--host-audio"; do for n in 64 1; do
timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --streams $n --steps 24 --warmup 6 $f 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$n streams $f', j['ms_per_step'], j['value'], j['audio'])"; done; done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
