# round 6, closing call: (1) VERDICT r05 item 6's counter pair -- FETCH_SIZE and SQ_WAIT_INST_ANY of the cfg-5 particle
# kernel on the particles as generated and sorted by 8 x 8-cell tile (experiments/particles_sorted_ab.py: 14 launches
# per ordering, in that order); (2) the fuzz campaign and (3) the multi-device soak on the round's final code
TAG=${1:-r06c}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE SQ_WAIT_INST_ANY; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/sorted_$c -- python3 $R/experiments/particles_sorted_ab.py > $O/sorted_$c.log 2>&1
done
python3 - $O <<'PY' > $O/sorted_pmc.txt
import csv, glob, sys
O = sys.argv[1]
names = ["as generated", "sorted by 8x8-cell tile", "sorted by tile, then by heading"]
for c in ("FETCH_SIZE", "SQ_WAIT_INST_ANY"):
    rows = []
    for f in glob.glob("%s/sorted_%s/**/*counter_collection.csv" % (O, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if "score_poses_compact_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals = [v for _, v in rows]
    per = len(vals) // 3
    for i, n in enumerate(names):
        part = vals[i * per:(i + 1) * per][4:]
        if part:
            print("%-18s %-34s launches %d  mean %.6g  min %.6g  max %.6g" % (c, n, len(part), sum(part) / len(part), min(part), max(part)))
PY
cat $O/sorted_pmc.txt; tail -4 $O/sorted_FETCH_SIZE.log
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cd $R
python3 experiments/fuzz_r04.py general 700000 8000 > $O/fuzz_general.txt 2>&1; tail -3 $O/fuzz_general.txt
python3 experiments/fuzz_r04.py multi 700000 500 > $O/fuzz_multi.txt 2>&1; tail -3 $O/fuzz_multi.txt
python3 experiments/fuzz_r04.py large 700000 300 > $O/fuzz_large.txt 2>&1; tail -3 $O/fuzz_large.txt
timeout 400 python3 experiments/soak_r05.py 90 31 > $O/soak.txt 2>&1; tail -5 $O/soak.txt
