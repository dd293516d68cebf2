# SQ counters of the three-launch particle scoring (prepare / screen / drain), cfg given as $1 (3 or 5)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-3}
O=$R/gpurun_out/r05b/pmc_split$CFG
rm -rf $O && mkdir -p $O
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/experiments/split_prof.py split $CFG > $O/$name.log 2>&1
}
pass valu SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python3 - $O <<'PY'
import csv, glob, collections, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_poses_kernel)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[m.group(1)]["us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %.5g" % (c, sum(v) / len(v)))
PY
