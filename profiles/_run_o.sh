cd $GRAFT_REPO_ROOT
cp infinisst_amd/libinfinisst_hip.so /tmp/base.so; cp infinisst_amd/libinfinisst_hip_sc1.so /tmp/sc1.so
for rep in 1 2; do for v in base sc1; do cp /tmp/$v.so infinisst_amd/libinfinisst_hip.so
timeout 300 python3 bench.py --no-cpu-baseline --no-streams64 --no-roofline --steps 32 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v 1 stream', j['ms_per_step'], j['value'])"; done; done
