cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
bash profiles/collect_round_profiles.sh > gpurun_out/r02/collect.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02/pytest_gpu.log
tail -1 gpurun_out/r02/bench_v1.json | cut -c1-300
