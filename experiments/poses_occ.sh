# the particle kernel at five waves per SIMD (scalar exp constants free the registers): cfg-3 / cfg-5 kernel ms
for lib in "" experiments/bin/poses_w4s.so experiments/bin/poses_w5.so "" experiments/bin/poses_w5.so; do
  if [ -n "$lib" ]; then export NDT2D_HIP_LIB=$PWD/$lib; else unset NDT2D_HIP_LIB; fi
  echo "== lib ${lib:-in-tree}"
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-default-search --no-c-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['particle_filter']; c=d['cfg5_single_gpu']
print('cfg-3 kernel_ms %.4f (%s)  cfg-5 kernel_ms %.4f ms/step %.4f' % (p['kernel_ms'], p['variant'], c['kernel_ms'], c['ms_per_step']))"
done
