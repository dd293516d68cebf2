cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
python experiments/small_plan_sweep.py defaults 0,0 5,3 5,2 5,1 7,2 7,1 8,2 8,1 9,1 13,1 4,4 4,3 3,5 6,2 > gpurun_out/r06l/plan_sweep.txt 2>&1
cat gpurun_out/r06l/plan_sweep.txt | grep -v amdgpu.ids
python -m pytest tests -m gpu -q > gpurun_out/r06l/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed" gpurun_out/r06l/gpu_tests.log | tail -3
