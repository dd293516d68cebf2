# table-driven exp (-DNDT2D_EXP_TABLE) in the compacted large search: time and score differences
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab5
mkdir -p $O
run() { L=$1; shift; echo "== ${L:-current} $*" >> $O/t.txt; if [ -n "$L" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$L.so; else unset NDT2D_HIP_LIB; fi; python bench.py "$@" --no-cpu-baseline --no-particles --no-default-search --no-anchors 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['config']['kernel_variant'], d['match_result']['best_index'], repr(d['match_result']['score']))" >> $O/t.txt; }
for i in 1 2 3; do
run "" --steps 100 --warmup 5
run lane_exptab --steps 100 --warmup 5
done
cat > /tmp/cmp_scores.py <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from ndt_2d_amd import ScanMatcherNDT, synth
m = ScanMatcherNDT(0); m.initialize("x", **synth.matcher_params(2)); m.addScans(synth.map_scans(2))
g, p, _ = synth.query_scan(2)
r = m.matchScan(g, p, want_scores=True)
np.save(sys.argv[1], r["scores"]); print(m.last_variant(), r["best_index"], repr(r["score"]))
PY
unset NDT2D_HIP_LIB; python /tmp/cmp_scores.py /tmp/s0.npy >> $O/t.txt 2>&1
NDT2D_HIP_LIB=$PWD/experiments/bin/lane_exptab.so python /tmp/cmp_scores.py /tmp/s1.npy >> $O/t.txt 2>&1
python -c "import numpy as np; a=np.load('/tmp/s0.npy'); b=np.load('/tmp/s1.npy'); d=np.abs(a-b); print('max |dscore|', d.max(), 'rel to ulp of score', (d/np.spacing(np.abs(a)+1e-300)).max(), 'nonzero diffs', int((d>0).sum()), 'of', d.size)" >> $O/t.txt
cat $O/t.txt
