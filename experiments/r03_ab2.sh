# where the records live, on ONE workload (cfg-2): compacted in LDS with six waves per SIMD
# (default), the whole grid's in LDS with four, gathered from HBM with four and with six
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab2
mkdir -p $O
run() { echo "== $*" >> $O/records.txt; env "$@" python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-particles --no-default-search --no-anchors 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['config']['kernel_variant'], d['match_result']['best_index'])" >> $O/records.txt; }
for i in 1 2; do
run X=1
run NDT2D_LANE_COMPACT=0
run NDT2D_LANE_RECORDS=global NDT2D_LANE_GATHER6=0
run NDT2D_LANE_RECORDS=global
done
cat $O/records.txt
