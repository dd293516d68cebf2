# per-kernel durations of the large search (rocprofv3 kernel trace) for the in-tree library:
#     bash experiments/kt_lane.sh tag [NAME=VALUE ...]
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt_lane_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-particles --no-default-search > $O/kt.log 2>&1
echo "== $TAG $*"
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs cut -c1-60,200-300 | head -9
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-40s calls %5s avg %10.1f ns' % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])))
"
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
