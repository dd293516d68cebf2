cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06g; rm -rf $O; mkdir -p $O
BENCH="python3 $R/bench.py --steps 4 --warmup 2 --prewarm 0 --no-cpu-baseline --no-default-search --no-anchors --no-c-host --no-particles"
for d in 0 1; do
  export NDT2D_LANE_DEFER=$d
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/d$d -- $BENCH > $O/d$d.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('%s/d%d/**/*counter_collection.csv' % (O, d), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].split('::')[-1]
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in ('match_lane_compact_kernel', 'match_lane_compact_fused_kernel', 'match_lane_scores_kernel', 'match_reduce_kernel', 'match_reduce_final_kernel'):
        if k in agg:
            print('defer=%d %-34s' % (d, k), {c: sum(v[-4:]) / len(v[-4:]) for c, v in agg[k].items()})
PY
