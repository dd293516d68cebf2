for v in "" 2 4 8; do
  if [ -n "$v" ]; then export NDT2D_LANE_PARTS=$v; else unset NDT2D_LANE_PARTS; fi
  python experiments/lattice_ab.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parts=%-4s' % '${v:-auto}', ' '.join('%s %.4f/%.4f/%s' % (n, d[n]['kernel_ms'], d[n].get('call_ms', 0), d[n]['scores_sha'][:6]) for n in ('mid_6760','mid_23660','cfg2')))"
done
