cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
python experiments/small_plan_sweep.py defaults 0,0 5,3 5,1 > gpurun_out/r06m/plan.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06m/plan.txt
for i in 1 2; do ndt_2d_amd/ndt2d_latency_probe 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('match_scan_us','add_scans_us','mapper_cycle_us','pf_measure_500_particles_us')}, {k:d['real_lidar_map'][k] for k in ('match_scan_us','add_scans_us','mapper_cycle_us')})"; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_near_ties.py tests/test_gpu_fuzz.py tests/test_gpu_c_consumer_latency.py tests/test_gpu_bounded_poll.py -q -x > gpurun_out/r06m/tests.log 2>&1; echo rc=$?; grep -n "passed\|failed" gpurun_out/r06m/tests.log | tail -2
