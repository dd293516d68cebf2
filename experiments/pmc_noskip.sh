# The exact path alone: cfg-2 under `lane-noskip` (every unit evaluated exactly, 1.44e9 exact evaluations per
# launch = 2.25e7 per wave-lane), at four / six / eight waves per SIMD, with SQ counters.
#   bash experiments/build_variant_lib.sh lane_t512 ndt2d_match_lane.hip -DNDT2D_LANE_THREADS_COMPACT=512
#   bash experiments/build_variant_lib.sh lane_t1024b ndt2d_match_lane.hip -DNDT2D_LANE_THREADS_COMPACT=1024 -DNDT2D_EXP_SCALAR_CONSTANTS=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_noskip
rm -rf $O && mkdir -p $O
for lib in experiments/bin/lane_t512.so "" experiments/bin/lane_t1024b.so; do
  if [ -n "$lib" ]; then export NDT2D_HIP_LIB=$R/$lib; tag=$(basename $lib .so); else unset NDT2D_HIP_LIB; tag=intree; fi
  echo "== ${lib:-in-tree (768 threads)}"
  python3 $R/experiments/noskip_case.py 2>&1 | tail -1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/$tag -- python3 $R/experiments/noskip_case.py > $O/$tag.log 2>&1
  python3 - $O/$tag <<'PY'
import csv, glob, re, sys
d = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m and m.group(1).startswith("match_lane_compact"):
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
if d:
    print("  " + "  ".join("%s=%.4g" % kv for kv in sorted(d.items())))
    print("  issue fraction %.3f; VALU per exact evaluation (wave level) %.1f; LDS busy %.2f of CU cycles, conflicts %.2f of that; waits %.2f of wave life"
          % (d["SQ_INSTS_VALU"] / (d["us"] * 1e-6) / 614.4e9, d["SQ_INSTS_VALU"] / 2.25e7, d["SQ_LDS_IDX_ACTIVE"] / d["SQ_BUSY_CU_CYCLES"],
             d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]))
PY
done
