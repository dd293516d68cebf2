# the small-lattice search's beam chunks per tile (NDT2D_SMALL_CHUNKS; the plan's own choice = "plan")
for c in plan 4 6 8 12 16; do
  if [ "$c" != "plan" ]; then export NDT2D_SMALL_CHUNKS=$c; else unset NDT2D_SMALL_CHUNKS; fi
  python experiments/lattice_ab.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunks=%-5s' % '$c', ' '.join('%s %.4f/%.4f' % (n, d[n]['kernel_ms'], d[n].get('call_ms', 0)) for n in ('default','cfg1','mid_1352')))"
done
