# how sensitive is the large search to SCALAR instructions?  Variants with 16 / 32 idle s_mov_b32 (or 16 v_mov_b32)
# per look-up group (1.25 million groups per cfg-2 launch: + 2.0e7 / 4.0e7 on 8.7e7 SALU, + 2.0e7 on 2.37e8 VALU);
# built from a temporary patch of ndt2d_lane_fn.h (-DNDT2D_LANE_SALU_PAD=<n> / -DNDT2D_LANE_VALU_PAD=<n>)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
run() {  # name lib
  NDT2D_HIP_LIB=$2 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-default-search --no-anchors --no-c-host --no-particles --detail-file $PWD/$O/$1.json > /dev/null 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/$1.json')); print('%-10s' % '$1', 'ms/step %.4f' % d['ms_per_step'], 'kernel ms %.4f' % d['roofline']['kernel_ms_avg'], d['match_result']['best_index'])"
}
for rep in 1 2; do
  run product $PWD/ndt_2d_amd/libndt2d_hip.so
  for v in salu16 valu16 valu16e64 salu16lit; do run $v $PWD/experiments/bin/$v.so; done
done
