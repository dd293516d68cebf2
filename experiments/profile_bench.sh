cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof7
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-particles > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $O/sq1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles > $O/write.log 2>&1
cd $O
find . -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
for d in sq1 sq2 fetch write; do f=$(find $d -name "*counter_collection.csv" | head -1); python3 $R/experiments/pmc_summary.py $f > $O/pmc_$d.txt 2>&1; done
find . -name "*.csv" -size +2M -delete
head -5 $O/kernel_stats.csv; cat $O/pmc_sq1.txt $O/pmc_sq2.txt $O/pmc_fetch.txt $O/pmc_write.txt | grep -A12 match_lane
