cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in call_breakdown call_breakdown_trace; do
  rm -rf /tmp/pp; (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- $R/experiments/bin/$b) > /tmp/pp.log 2>&1
  echo "== $b"; tail -5 /tmp/pp.log | head -4
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "match_small" in r["Name"]:
        print("  ", r["Name"][:75].replace("ndt2d::(anonymous namespace)::",""), r["Calls"], "avg", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
done
