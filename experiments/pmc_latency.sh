# Latency-oriented SQ counters for the search kernel (average SMEM / LDS / ifetch latency).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lat
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_SMEM --output-format csv -d $O/a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/b.log 2>&1
for d in a b; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f | grep -A10 match_lane; done
find $O -name "*.csv" -size +1M -delete
