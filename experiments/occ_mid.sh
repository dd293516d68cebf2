# the large search at four / six / eight waves per SIMD (512 / 768 / 1024-thread blocks, two per CU; eight need
# exp's constants scalar to fit 64 registers): cfg-2 and the mid-size lattices
#   bash experiments/build_variant_lib.sh lane_t512 ndt2d_match_lane.hip -DNDT2D_LANE_THREADS_COMPACT=512
#   bash experiments/build_variant_lib.sh lane_t1024b ndt2d_match_lane.hip -DNDT2D_LANE_THREADS_COMPACT=1024 -DNDT2D_EXP_SCALAR_CONSTANTS=1
for lib in experiments/bin/lane_t512.so "" experiments/bin/lane_t1024b.so; do
  if [ -n "$lib" ]; then export NDT2D_HIP_LIB=$PWD/$lib; else unset NDT2D_HIP_LIB; fi
  echo "== lib ${lib:-in-tree (768 threads, six waves per SIMD)}"
  timeout 120 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-default-search --no-c-host --no-particles --no-anchors 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg-2: ms/step %.4f kernel_ms %.4f' % (d['ms_per_step'], r['kernel_ms_avg']))"
  timeout 100 python experiments/mid_lattice_parts.py 2>&1 | grep -E "items  (3549|6760|13520|23660)" | sed -e 's/  small.*auto/ auto/'
done
