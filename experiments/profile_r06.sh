# Round-6 profile of the default bench: rocprofv3 kernel trace + stats, the PMC counters in
# separate passes (SQ: 8 counters per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass), as
# MI355X_MICROARCH.md prescribes -- and, new, one launch per 1-of-8 share of the two 8-GPU
# workloads (bench.py --profile-shares) so that a `--gpus N` line has counters for each rank's
# share.  --pmc is only ever combined with --kernel-trace.  Through gpurun, from the repo root:
#     bash experiments/profile_r06.sh [tag]
# leaves gpurun_out/prof_<tag>/{kernel_stats.csv,pmc.json,pmc_summary.txt,bench.json,...}.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
BENCH="python3 $R/bench.py --steps 4 --warmup 2 --prewarm 0 --no-cpu-baseline --no-default-search --no-anchors --no-c-host"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-default-search --no-anchors --no-c-host > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $BENCH > $O/$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32
pass sq4 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 $R/experiments/pmc_to_json.py $O > $O/pmc_summary.txt 2> $O/pmc_to_json.err
# the 8-GPU workloads share by share on this one GPU
SH="python3 $R/bench.py --profile-shares"
spass() {
  name=$1; shift
  mkdir -p $O/shares
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/shares/$name -- $SH > $O/shares_$name.log 2>&1
}
spass sq1 SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY
spass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32
spass sq3 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32
spass fetch FETCH_SIZE
spass write WRITE_SIZE
python3 $R/experiments/pmc_shares.py $O $(python3 $R/bench.py --print-source-hash) > $O/shares_summary.txt 2>> $O/pmc_to_json.err
# this session's counters in place (bench.py reads profiles/r06_pmc.json), then what prices them: the hot kernels'
# static in-loop mix (experiments/asm_loop_mix.py) and -- new in round 6 -- the headline kernel's DYNAMIC mix: path
# counts of a -DNDT2D_LANE_PATHS build (experiments/bin/lane_paths.so, built by experiments/build_variant_lib.sh
# before the call) x the paths' instruction lists, checked class by class against the counters just taken
cp $O/pmc.json $R/profiles/r06_pmc.json
python3 $R/experiments/asm_loop_mix.py > $R/profiles/r06_valu_mix.json 2> $O/asm_loop_mix.err
NDT2D_HIP_LIB=$R/experiments/bin/lane_paths.so python3 $R/experiments/lane_paths.py 2 > $R/profiles/r06_lane_paths.json 2> $O/lane_paths.err
python3 $R/experiments/lane_path_mix.py $R/profiles/r06_lane_paths.json $R/profiles/r06_pmc.json > $R/profiles/r06_lane_path_mix.json 2> $O/lane_path_mix.err
cp $R/profiles/r06_valu_mix.json $R/profiles/r06_lane_paths.json $R/profiles/r06_lane_path_mix.json $O/
# the plain bench lines LAST: stdout = the short line the driver parses, --detail-file = the full record
python3 $R/bench.py --detail-file $O/bench_detail.json > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --steps 20 --warmup 5 --detail-file $O/bench_driver_flags_detail.json > $O/bench_driver_flags.json 2>> $O/bench.err
NDT2D_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 8 --steps 4 --warmup 2 --detail-file $O/bench_8ranks_one_gpu_gloo_detail.json > $O/bench_8ranks_one_gpu_gloo.json 2>> $O/bench.err
wc -c $O/bench.json $O/bench_driver_flags.json $O/bench_8ranks_one_gpu_gloo.json
# the multi-device matcher through the plain-C probe (dealing overhead, fan-out, cfg-5 sharded by default)
bash $R/experiments/profile_multi_device_r06.sh > $O/multi_device.log 2>&1
cp $R/gpurun_out/r06/multi_device/summary.json $O/multi_device_summary.json
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
cut -c1-150 $O/kernel_stats.csv | head -8
head -60 $O/pmc_summary.txt
cat $O/shares_summary.txt
