# cycle breakdown (experiments/cycle_breakdown.c) + a GPU timeline of the last cycles (kernel + memory-copy trace)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/cycle; mkdir -p $O experiments/bin
gcc -O2 -std=c99 -I include experiments/cycle_breakdown.c -L ndt_2d_amd -lndt2d_hip -lm -Wl,-rpath,$PWD/ndt_2d_amd -o experiments/bin/cycle_breakdown
for i in 1 2; do experiments/bin/cycle_breakdown 2000 0; done > $O/real.txt 2>&1
for i in 1 2; do experiments/bin/cycle_breakdown 2000 1; done > $O/toy.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace -o t -- $R/experiments/bin/cycle_breakdown 100 ${1:-0} > $R/$O/trace.log 2>&1 || true
cd $R
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/cycle/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', ''))))
for f in glob.glob('gpurun_out/cycle/trace/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
tail = rows[-24:]
t0 = tail[0][0]
with open('gpurun_out/cycle/timeline.txt', 'w') as o:
    for s, e, n in tail:
        o.write('%9.2f %9.2f %7.2f  %s\n' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $O/trace
