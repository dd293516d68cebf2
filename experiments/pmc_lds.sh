# LDS counters of the large search: bank conflicts and LDS-busy cycles, cfg-2 (1.0 0.5) and a mid-size lattice (1.0 0.1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lds
rm -rf $O && mkdir -p $O
for c in "1.0 0.5" "1.0 0.1"; do
  t=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --output-format csv -d $O/lds_$t -- python3 $R/experiments/mid_lattice_case.py $c > $O/lds_$t.log 2>&1
  echo "== $c"
  python3 - $O/lds_$t <<'PY'
import csv, glob, collections, re, sys
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m and m.group(1).startswith("match_lane_compact"):
            agg[m.group(1)][r["Counter_Name"]] = float(r["Counter_Value"])
            agg[m.group(1)]["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    for k, d in agg.items():
        print(" ", k, " ".join("%s=%.4g" % kv for kv in sorted(d.items())))
PY
done
tail -3 $O/lds_1.0_0.1.log
