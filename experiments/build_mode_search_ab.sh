# the searches of lattice_ab.py on a host-built grid (compacted records in LDS) against the same
# grid built on the device (what addScans of >= 32,768 points -- a loop closure -- produces)
for mode in "" device "" device; do
  if [ -n "$mode" ]; then export NDT2D_AB_BUILD_MODE=$mode; else unset NDT2D_AB_BUILD_MODE; fi
  python experiments/lattice_ab.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-7s' % '${mode:-host}', ' | '.join('%s %.4f %s' % (n, d[n]['kernel_ms'], d[n]['variant'].split('/',2)[-1][:34]) for n in ('default','cfg1','mid_1352','mid_6760','mid_23660','cfg2')))"
done
