# SQ counters of the large-lattice search kernel for the in-tree library; optional
# environment assignments as arguments are exported first (A/B of env knobs):
#     bash experiments/pmc_lane.sh tag [NAME=VALUE ...]
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lane_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $O/a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles --no-default-search > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM --output-format csv -d $O/b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-particles --no-default-search > $O/b.log 2>&1
echo "== $TAG $*"
for p in a b; do
  f=$(find $O/$p -name "*counter_collection.csv" | head -1)
  python3 $R/experiments/pmc_summary.py $f | grep -A10 "match_lane"
done
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
