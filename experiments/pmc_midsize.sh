# Mid-size lattices: per-kernel time, the gaps between the kernels of one
# search, and the SQ counters of the search kernel, for three lattices on cfg-2's map and scan.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_midsize
rm -rf $O && mkdir -p $O
for c in "1.0 0.02" "1.0 0.1" "1.0 0.35" "1.0 0.5"; do
  t=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$t -- python3 $R/experiments/mid_lattice_case.py $c > $O/kt_$t.log 2>&1
  echo "== $c: $(tail -1 $O/kt_$t.log)"
  python3 $R/experiments/kernel_gaps.py $O/kt_$t
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_$t -- python3 $R/experiments/mid_lattice_case.py $c > $O/sq_$t.log 2>&1
  python3 - $O/sq_$t <<'PY'
import csv, glob, collections, sys
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        if "match" in k or "outer" in k:
            agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3, r["Grid_Size"], r["Workgroup_Size"]))
    for k, c in agg.items():
        for name, vals in sorted(c.items()):
            v = vals[-1]
            print("  pmc", k[:40], name, "%.5g" % v[0], "(%.1f us, grid %s wg %s)" % (v[1], v[2], v[3]))
PY
done
