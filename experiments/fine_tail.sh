# beam parts with the launch's last items handed out chunk by chunk (NDT2D_LANE_FINE_ITEMS: how many items; 0: none)
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "part or slab or mid" 2>&1 | tail -2
for f in 0 256 512 1024 2048 4096; do
  echo "== NDT2D_LANE_FINE_ITEMS=$f"
  NDT2D_LANE_FINE_ITEMS=$f timeout 100 python experiments/mid_lattice_parts.py 2>&1 | grep -E "items  (3549|6760|13520)" | sed -e 's/  small.*auto/ auto/'
done
echo "== default"
timeout 100 python experiments/mid_lattice_parts.py 2>&1 | grep -E "items  (3549|3920|6760|13520)" | sed -e 's/  small.*auto/ auto/'
