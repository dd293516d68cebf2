# The VALU issue table behind bench.py's roofline (profiles/r05_ubench_issue.json), through
# gpurun from the repo root:   bash experiments/ubench_issue.sh
#   1. experiments/bin/ubench_issue: every VALU instruction class of the two hot kernels, timed,
#   2. the SQ_INSTS_VALU_* class counters for every one of those loops: which class counts which
#      instruction (and: does any counter tell a 2-cycle instruction from a 4-cycle one?)
#      (--pmc only ever with --kernel-trace; the program itself after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/ubench
rm -rf $O && mkdir -p $O
$R/experiments/bin/ubench_issue > $O/ubench_issue.jsonl 2> $O/ubench_issue.err
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
grep -oE "\bSQ_[A-Z0-9_]+" $O/list_avail.txt | sort -u | tr '\n' ' ' > $O/sq_counters.txt
# which SQ_INSTS_VALU_* class counts which instruction: every loop once (4 waves per SIMD) under
# the class counters, two passes; the timed dispatch of kernel k<OP> is the larger of its two
for pass in a b; do
  if [ $pass = a ]; then C="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT";
  else C="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/cls_$pass -- $R/experiments/bin/ubench_issue -1 4 > $O/cls_$pass.jsonl 2> $O/cls_$pass.err
done
python3 - $O > $O/pmc_classes.json 2>> $O/cls_a.err <<'PY'
import collections, csv, glob, json, re, sys
O = sys.argv[1]
names = {}
for line in open(O + "/ubench_issue.jsonl"):
    d = json.loads(line)
    if "op" in d:
        names[d["op"]] = d["instruction"]
best = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ("a", "b"):
    for f in glob.glob(O + "/cls_%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k<(\d+)>", r["Kernel_Name"])
            if m:
                op = int(m.group(1))
                best[op][r["Counter_Name"]] = max(best[op][r["Counter_Name"]], float(r["Counter_Value"]))
out = {}
for op in sorted(best):
    c = best[op]
    total = c.get("SQ_INSTS_VALU", 0.0)
    row = {"SQ_INSTS_VALU": total}
    for k, v in sorted(c.items()):
        if k.startswith("SQ_INSTS_VALU_") and total > 0 and v / total > 0.01:
            row[k.replace("SQ_INSTS_VALU_", "")] = round(v / total, 3)
    if total > 0 and "SQ_BUSY_CU_CYCLES" in c:
        row["busy_cu_cycles_per_inst_x1024"] = round(c["SQ_BUSY_CU_CYCLES"] / total * 4, 3)
    out[names.get(op, str(op))] = row
json.dump(out, sys.stdout, indent=1)
PY
find $O -name "*.db" -delete
find $O -name "*.csv" -size +1M -delete
tail -5 $O/ubench_issue.jsonl
head -c 3000 $O/pmc_classes.json
