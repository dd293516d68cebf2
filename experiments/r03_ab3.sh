# items none of whose candidates scored: skip the wave reductions (-DNDT2D_LANE_ZERO_ITEMS), cfg-2 and cfg-4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab3
mkdir -p $O
run() { L=$1; shift; echo "== ${L:-current} $*" >> $O/zero.txt; if [ -n "$L" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$L.so; else unset NDT2D_HIP_LIB; fi; python bench.py "$@" --no-cpu-baseline --no-particles --no-default-search --no-anchors 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['config']['kernel_variant'], d['match_result']['best_index'])" >> $O/zero.txt; }
for i in 1 2 3; do
run "" --steps 100 --warmup 5
run lane_zero --steps 100 --warmup 5
done
for i in 1 2; do
run "" --workload cfg4 --steps 5 --warmup 1 --prewarm 0.2
run lane_zero --workload cfg4 --steps 5 --warmup 1 --prewarm 0.2
done
cat $O/zero.txt
