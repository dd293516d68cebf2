# the large search without its exact evaluations (wrong results; timing only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab4
mkdir -p $O
run() { L=$1; shift; echo "== ${L:-current} $*" >> $O/t.txt; if [ -n "$L" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$L.so; else unset NDT2D_HIP_LIB; fi; python bench.py "$@" --no-cpu-baseline --no-particles --no-default-search --no-anchors 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['config']['kernel_variant'], d['match_result']['best_index'])" >> $O/t.txt; }
for i in 1 2; do
run "" --steps 100 --warmup 5
run lane_noexact --steps 100 --warmup 5
NDT2D_LANE_PRETEST=0 run lane_noexact --steps 100 --warmup 5
done
cat $O/t.txt
