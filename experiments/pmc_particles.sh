# SQ counters for the particle scoring kernel (cfg-3), screened vs exact phase A.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pp
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY --output-format csv -d $O/a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/a.log 2>&1
f=$(find $O/a -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f | grep -A9 "score_poses_compact"
find $O -name "*.csv" -size +1M -delete
