# SQ counters of the small-lattice search on one workload ($1: defaults | d720 | cfg1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_small_$1
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- python3 $R/experiments/small_plan_sweep.py $1 0,0 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $O/b -- python3 $R/experiments/small_plan_sweep.py $1 0,0 > $O/b.log 2>&1
for d in a b; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f | grep -A12 "match_small"; done
find $O -name "*.csv" -size +1M -delete
