# Round-3 profile of the default bench: rocprofv3 kernel trace + stats, then the PMC
# counters in separate passes (SQ: 8 counters per pass; FETCH_SIZE and WRITE_SIZE cannot
# share a pass), as MI355X_MICROARCH.md prescribes.  --pmc is only ever combined with
# --kernel-trace.  Usage (through gpurun, from the repository root):
#     bash experiments/profile_r03.sh [tag]
# leaves gpurun_out/prof_<tag>/{kernel_stats.csv,pmc.json,pmc_summary.txt,bench.json}.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
BENCH="python3 $R/bench.py --steps 4 --warmup 2 --prewarm 0 --no-cpu-baseline --no-default-search --no-anchors"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-default-search --no-anchors > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $BENCH > $O/$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 $R/experiments/pmc_to_json.py $O > $O/pmc_summary.txt 2> $O/pmc_to_json.err
# the plain bench line LAST, with this session's counters in place (bench.py reads profiles/r03_pmc.json)
cp $O/pmc.json $R/profiles/r03_pmc.json
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/bench.err
# the 8-GPU lattice (cfg-4) on this one GPU: where do its instructions go?
C4="python3 $R/bench.py --workload cfg4 --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-default-search --no-anchors --no-particles"
mkdir -p $O/cfg4
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $O/cfg4/sq1 -- $C4 > $O/cfg4/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/cfg4/sq2 -- $C4 > $O/cfg4/sq2.log 2>&1
python3 $R/experiments/pmc_to_json.py $O/cfg4 > $O/cfg4_pmc_summary.txt 2>> $O/pmc_to_json.err
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
cut -c1-150 $O/kernel_stats.csv | head -8
cat $O/pmc_summary.txt | head -80
