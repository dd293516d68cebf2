# the large search with its epilogue fused into every work item (NDT2D_LANE_DEFER=0) against the round-6 form
# (the search kernel stores an item's 64 sums, match_lane_scores_kernel does the rest): step and kernel time at cfg-2,
# cfg-4 on one GPU, the result; then the GPU tests that pin the search
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
for d in 0 1 0 1; do
  NDT2D_LANE_DEFER=$d python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-default-search --no-anchors --no-c-host --no-particles --detail-file $PWD/$O/cfg2_defer$d.json > /dev/null 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/cfg2_defer$d.json')); print('cfg-2 defer=$d', 'ms/step %.4f' % d['ms_per_step'], 'kernel ms %.4f' % d['roofline']['kernel_ms_avg'], d['match_result']['best_index'], d['match_result']['score'], d['config']['kernel_variant'])"
done
for d in 0 1; do
  NDT2D_LANE_DEFER=$d python3 experiments/cfg4_step.py
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_near_ties.py tests/test_gpu_multi_device.py tests/test_gpu_fuzz.py tests/test_gpu_overflow_cell.py tests/test_gpu_dist_sharded.py -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed\|Error" $O/gpu_tests.log | tail -5
