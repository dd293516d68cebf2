# A/B of builds of the particle kernel (experiments/bin/<name>.so from build_variant_lib.sh; "" = in-tree)
for lib in "$@"; do
  if [ "$lib" != "in-tree" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$lib.so; else unset NDT2D_HIP_LIB; fi
  echo "== lib $lib"
  timeout 200 python experiments/particles_ab.py 2>/dev/null
done
