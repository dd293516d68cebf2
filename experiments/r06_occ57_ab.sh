# the large search at five / seven waves per SIMD (round 5's occ_mid.sh had four / six / eight).  A block's waves go
# round the four SIMDs, so 640- / 896-thread blocks (10 / 14 waves) leave no room for a second block on a CU
# (lane640 / lane896: 0.864 / 0.728 ms against 0.459); four blocks of 5 / 6 / 7 waves per CU (lane320x4 / lane384x4 /
# lane448x4: build_variant_lib.sh with -DNDT2D_LANE_THREADS_COMPACT=<n> and, from a two-line patch that makes them
# macros, -DNDT2D_LANE_COMPACT_WAVES_PER_EU=<w> -DNDT2D_LANE_BLOCKS_PER_CU=4) do not fit either -- the map and the
# records are ~70 KB of LDS per block, two blocks per CU: 0.88 / 0.80 / 0.75 ms.  Five and seven waves per SIMD cannot
# be had: a CU holds two blocks, and a block's waves must be a multiple of four to sit evenly.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
run() {  # name lib
  NDT2D_HIP_LIB=$2 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-default-search --no-anchors --no-c-host --no-particles --detail-file $PWD/$O/$1.json > /dev/null 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/$1.json')); print('%-10s' % '$1', 'ms/step %.4f' % d['ms_per_step'], 'kernel ms %.4f' % d['roofline']['kernel_ms_avg'], d['match_result']['best_index'], d['match_result']['score'])"
}
for rep in 1 2; do
  run product $PWD/ndt_2d_amd/libndt2d_hip.so
  for v in lane320x4 lane384x4 lane448x4; do run $v $PWD/experiments/bin/$v.so; done
done
