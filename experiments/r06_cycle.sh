# round 6: the mapper's cycle stage by stage (experiments/cycle_breakdown.c) on both maps, the latency probe, and
# a GPU timeline of the last cycles under rocprofv3 --kernel-trace (no counters)
cd $GRAFT_REPO_ROOT
TAG=${1:-r06b}
O=gpurun_out/$TAG; mkdir -p $O
R=$PWD
for t in 1 0; do experiments/bin/host_build_phases $t 0; done > $O/host_build_phases.txt 2>&1
experiments/bin/cycle_breakdown 2000 1 > $O/cycle_toy.txt 2>&1
experiments/bin/cycle_breakdown 2000 0 > $O/cycle_big.txt 2>&1
ndt_2d_amd/ndt2d_latency_probe > $O/probe.json 2> $O/probe.err
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace$t -o t -- $R/experiments/bin/cycle_breakdown 100 $t > $R/$O/trace$t.log 2>&1 || true
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for t in (1, 0):
    rows = []
    for f in glob.glob('%s/trace%d/**/*kernel_trace.csv' % (O, t), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            short = name.split('(')[0].split('::')[-1][:40]
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', ''))))
    rows.sort()
    tail = rows[-16:]
    if not tail:
        continue
    t0 = tail[0][0]
    with open('%s/timeline_%s.txt' % (O, 'toy' if t else 'big'), 'w') as o:
        for s, e, n in tail:
            o.write('%9.2f %9.2f %7.2f  %s\n' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $O/trace1 $O/trace0
cat $O/host_build_phases.txt $O/cycle_toy.txt $O/cycle_big.txt $O/timeline_toy.txt $O/timeline_big.txt; cat $O/probe.json
