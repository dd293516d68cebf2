# SQ counters of the search kernel for the library given as $1 (experiments/bin/$1.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export NDT2D_HIP_LIB=$R/experiments/bin/$1.so
O=$R/gpurun_out/pmc_$1
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $O/a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/a.log 2>&1
f=$(find $O/a -name "*counter_collection.csv" | head -1); echo "== $1"; python3 $R/experiments/pmc_summary.py $f | grep -A9 "match_lane"
find $O -name "*.csv" -size +1M -delete
