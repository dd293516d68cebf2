for lib in "" screen_interleave "" screen_interleave; do
  if [ -n "$lib" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$lib.so; else unset NDT2D_HIP_LIB; fi
  echo "== lib ${lib:-in-tree}"
  timeout 200 python experiments/particles_ab.py 2>/dev/null
done
