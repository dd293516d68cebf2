# SQ counters for the particle scoring kernel at cfg-3 and cfg-5 (experiments/particles_ab.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pp5
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY --output-format csv -d $O/a -- python3 $R/experiments/particles_ab.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD TA_BUSY_avr --output-format csv -d $O/b -- python3 $R/experiments/particles_ab.py > $O/b.log 2>&1
for d in a b; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f | grep -A9 "score_poses_compact"; done
find $O -name "*.csv" -size +1M -delete
