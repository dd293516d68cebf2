# rocprofv3 passes for the state at the end of round 1 (particle kernel with the packed
# FP32 screen): kernel trace + stats of the default bench (search + particles), and the
# SQ counters of the particle kernel.  Run from the repository root through gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof12
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/sq.log 2>&1
cd $O
find kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
f=$(find sq -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f > $O/pmc_sq.txt 2>&1
find . -name "*.csv" -size +2M -delete
cut -c1-160 $O/kernel_stats.csv | head -12; grep -A9 "score_poses_compact\|match_lane" $O/pmc_sq.txt | head -40
