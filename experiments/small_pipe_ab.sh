# A/B of the small-lattice search with / without the pipelined exact evaluations
# (experiments/bin/small_nopipe.so: -DNDT2D_SMALL_NO_PIPELINE): kernel ms / call ms / scores hash
for i in 1 2; do
  for lib in in-tree small_nopipe; do
    if [ "$lib" != "in-tree" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$lib.so; else unset NDT2D_HIP_LIB; fi
    python experiments/lattice_ab.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-13s' % '$lib', ' '.join('%s %.4f/%.4f/%s' % (n, d[n]['kernel_ms'], d[n].get('call_ms', 0), d[n]['scores_sha'][:6]) for n in ('default','cfg1','mid_1352','mid_6760')))"
  done
done
