for lib in experiments/bin/lane_u4.so "" experiments/bin/lane_u16.so; do
  if [ -n "$lib" ]; then export NDT2D_HIP_LIB=$PWD/$lib; else unset NDT2D_HIP_LIB; fi
  echo "== lib ${lib:-in-tree (8 beams per group)}"
  timeout 120 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-default-search --no-c-host --no-particles --no-anchors 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg-2: ms/step %.4f kernel_ms %.4f' % (d['ms_per_step'], r['kernel_ms_avg']))"
  timeout 100 python experiments/mid_lattice_parts.py 2>&1 | grep -E "items  (3549|6760|23660)" | sed -e 's/  small.*auto/ auto/'
done
