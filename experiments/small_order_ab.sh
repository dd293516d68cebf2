for i in 1 2; do
  python experiments/lattice_ab.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('centre-first', ' '.join('%s %.4f/%s' % (n, d[n]['kernel_ms'], d[n]['scores_sha'][:6]) for n in ('default','cfg1','mid_1352','mid_6760')))"
  NDT2D_SMALL_CENTRE_FIRST=0 python experiments/lattice_ab.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('row-major   ', ' '.join('%s %.4f/%s' % (n, d[n]['kernel_ms'], d[n]['scores_sha'][:6]) for n in ('default','cfg1','mid_1352','mid_6760')))"
done
