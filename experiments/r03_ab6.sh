# the default search with and without the beam-local record cache: kernel durations (rocprofv3) and call latencies
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ab6
rm -rf $O && mkdir -p $O
for c in 1 0 1 0; do
  export NDT2D_SMALL_CACHE=$c
  echo "== NDT2D_SMALL_CACHE=$c" >> $O/t.txt
  $R/ndt_2d_amd/ndt2d_latency_probe >> $O/t.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$c -- $R/ndt_2d_amd/ndt2d_latency_probe > $O/p$c.log 2>&1
  python3 - <<PY >> $O/t.txt
import csv,glob
f=glob.glob("$O/p$c/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "match_small" in r["Name"] or "score_few" in r["Name"]:
        print("  ", r["Name"][:70].replace("ndt2d::(anonymous namespace)::",""), r["Calls"], "avg", r["AverageNs"], "min", r["MinNs"])
PY
  find $O -name "*.csv" -size +1M -delete
  rm -rf $O/p$c
done
cat $O/t.txt | cut -c1-330
