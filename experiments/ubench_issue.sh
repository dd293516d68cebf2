# The VALU issue table behind bench.py's roofline (profiles/r05_ubench_issue.json), through
# gpurun from the repo root:   bash experiments/ubench_issue.sh
#   1. experiments/bin/ubench_issue: every VALU instruction class of the two hot kernels, timed,
#   2. the SQ counters for four of those loops (FP64 fma / v_mov_b32 / v_add_u32 / v_cndmask):
#      do SQ_ACTIVE_INST_VALU or SQ_BUSY_CU_CYCLES tell a 2-cycle instruction from a 4-cycle one?
#      (--pmc only ever with --kernel-trace; the program itself after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/ubench
rm -rf $O && mkdir -p $O
$R/experiments/bin/ubench_issue > $O/ubench_issue.jsonl 2> $O/ubench_issue.err
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
grep -oE "\bSQ_[A-Z0-9_]+" $O/list_avail.txt | sort -u | tr '\n' ' ' > $O/sq_counters.txt
for op in 2 18 26 16 17 24; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU \
    --output-format csv -d $O/pmc_$op -- $R/experiments/bin/ubench_issue $op 4 > $O/pmc_$op.log 2>&1
  f=$(find $O/pmc_$op -name "*counter_collection.csv" | head -1)
  echo "== op $op" >> $O/pmc_calibration.txt
  python3 - $f >> $O/pmc_calibration.txt 2>&1 <<'PY'
import collections, csv, sys
# the timed launch is the larger of the kernel's two dispatches (10 / 1000 iterations)
best = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("void k<") or r["Kernel_Name"].startswith("k<"):
        best[r["Counter_Name"]] = max(best[r["Counter_Name"]], float(r["Counter_Value"]))
for c in sorted(best):
    print("   %-24s %.6g" % (c, best[c]))
PY
done
find $O -name "*.db" -delete
find $O -name "*.csv" -size +1M -delete
tail -5 $O/ubench_issue.jsonl
cat $O/pmc_calibration.txt | head -120
