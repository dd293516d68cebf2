# Mid-size lattices: per-kernel time, the gaps between the kernels of one
# search, and the SQ counters of the search kernel, for three lattices on cfg-2's map and scan.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/pmc_midsize
rm -rf $O && mkdir -p $O
for c in "1.0 0.02" "1.0 0.1" "1.0 0.35" "1.0 0.5"; do
  t=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_$t -- python3 $R/experiments/mid_lattice_case.py $c > $O/kt_$t.log 2>&1
  echo "== $c: $(tail -1 $O/kt_$t.log)"
  python3 $R/experiments/kernel_gaps.py $O/kt_$t
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_$t -- python3 $R/experiments/mid_lattice_case.py $c > $O/sq_$t.log 2>&1
  python3 - $O/sq_$t <<'PY'
import csv, glob, collections, re, sys
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        if "match" in k or "outer" in k:
            agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3, r["Grid_Size"], r["Workgroup_Size"]))
    for k, c in agg.items():
        d = {n: v[-1][0] for n, v in c.items()}
        v = list(c.values())[0][-1]
        us = v[1]
        print("  pmc %-34s %.1f us (counters on), grid %s wg %s: SQ_INSTS_VALU %.4g, SQ_BUSY_CU_CYCLES %.4g, SQ_WAVES %g, SQ_WAVE_CYCLES %.4g -> issue fraction %.3f of 614.4 G/s, waves alive %.2f of the launch"
              % (k, us, v[2], v[3], d.get("SQ_INSTS_VALU", 0), d.get("SQ_BUSY_CU_CYCLES", 0), d.get("SQ_WAVES", 0), d.get("SQ_WAVE_CYCLES", 0),
                 d.get("SQ_INSTS_VALU", 0) / (us * 1e-6) / 614.4e9,
                 4.0 * d.get("SQ_WAVE_CYCLES", 0) / max(1.0, d.get("SQ_WAVES", 1) * us * 1e-6 * 2.4e9)))
PY
done
