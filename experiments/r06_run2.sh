cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
python -m pytest tests -m gpu -x -q > gpurun_out/r06b/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -5 gpurun_out/r06b/gpu_tests.log
bash experiments/r06_cycle.sh r06b
