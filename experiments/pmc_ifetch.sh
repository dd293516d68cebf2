# Instruction-fetch side of the search kernel: SQ fetch requests and the instruction cache.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ifetch
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $O/a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $O/b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/b.log 2>&1
for d in a b; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f | grep -A9 "match_lane"; done
find $O -name "*.csv" -size +1M -delete
