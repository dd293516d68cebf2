cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06j
for w in defaults real30; do
  NDT2D_HIP_LIB=$PWD/experiments/bin/trace.so python experiments/small_trace.py $w isolated > gpurun_out/r06j/small_trace_$w.txt 2>&1
  grep -E "first block start|per wave" gpurun_out/r06j/small_trace_$w.txt
done
ndt_2d_amd/ndt2d_latency_probe > gpurun_out/r06j/probe.json 2>/dev/null; cat gpurun_out/r06j/probe.json
python -m pytest tests/test_gpu_bounded_poll.py tests/test_gpu_parity.py tests/test_gpu_particle_filter.py tests/test_gpu_single_pose_host.py tests/test_gpu_c_consumer_latency.py tests/test_gpu_near_ties.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
