# FETCH_SIZE / WRITE_SIZE of the search kernel for the library NDT2D_HIP_LIB points at.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetch_$1
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/w.log 2>&1
for d in f w; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f | grep -A1 "match_lane\|outer_table"; done
find $O -name "*.csv" -size +1M -delete
