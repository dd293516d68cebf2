# Instruction cache, scalar data cache, LDS / scalar wait counters of the large search: cfg-2 (1.0 0.5) and a
# mid-size lattice (1.0 0.1) -- what, besides VALU issue, the waves of the heavy regime wait for
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_frontend
rm -rf $O && mkdir -p $O
for c in "1.0 0.5" "1.0 0.1"; do
  t=$(echo $c | tr ' ' '_')
  n=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
             "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQC_TC_STALL" \
             "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA"; do
    n=$((n+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p${n}_$t -- python3 $R/experiments/mid_lattice_case.py $c > $O/p${n}_$t.log 2>&1
  done
  echo "== $c"
  python3 - $O $t <<'PY'
import csv, glob, collections, re, sys
O, t = sys.argv[1], sys.argv[2]
d = {}
for n in (1, 2, 3):
    for path in glob.glob("%s/p%d_%s/**/*counter_collection.csv" % (O, n, t), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if m and m.group(1).startswith("match_lane_compact"):
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["us_p%d" % n] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
print("  " + "  ".join("%s=%.4g" % kv for kv in sorted(d.items())))
PY
done
