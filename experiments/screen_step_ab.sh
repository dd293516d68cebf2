# the particle kernel's screening pass with 2 / 4 (in tree) / 8 / 16 beams in flight per step: cfg-3 / cfg-5 kernel ms
#   for v in 2 8 16; do bash experiments/build_variant_lib.sh screen_step$v ndt2d_poses_compact.hip -DNDT2D_SCREEN_STEP=$v; done
for lib in "" screen_step2 screen_step8 screen_step16 "" screen_step8; do
  if [ -n "$lib" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$lib.so; else unset NDT2D_HIP_LIB; fi
  echo "== lib ${lib:-in-tree}"
  timeout 200 python experiments/particles_ab.py 2>/dev/null
done
