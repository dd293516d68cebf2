# issue priority graded by the chunk's flagged beams (build variants prio_<hi>_<mid>) against the product, cfg-2,
# with the epilogue fused (NDT2D_LANE_DEFER=0) and as a kernel of its own (1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
run() {  # name lib defer
  NDT2D_HIP_LIB=$2 NDT2D_LANE_DEFER=$3 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-default-search --no-anchors --no-c-host --no-particles --detail-file $PWD/$O/$1_d$3.json > /dev/null 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/$1_d$3.json')); print('%-12s defer=$3' % '$1', 'ms/step %.4f' % d['ms_per_step'], 'kernel ms %.4f' % d['roofline']['kernel_ms_avg'], d['match_result']['best_index'], d['match_result']['score'])"
}
for d in 0 1; do
  run product $PWD/ndt_2d_amd/libndt2d_hip.so $d
  for v in 48_16 56_24 60_32 64_48; do run prio_$v $PWD/experiments/bin/prio_$v.so $d; done
  run product $PWD/ndt_2d_amd/libndt2d_hip.so $d
done
