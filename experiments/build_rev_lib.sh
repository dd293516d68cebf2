# experiments/bin/<name>.so: the library with ONE translation unit taken from a git revision
#   bash experiments/build_rev_lib.sh <name> <file.hip> <rev> [flags...]
# (the headers are today's: use only where the unit still compiles against them)
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; FILE=$2; REV=$3; shift 3
D=$R/experiments/bin/obj_$NAME
mkdir -p $D
git -C $R show $REV:ndt_2d_amd/csrc/$FILE > $D/$FILE
OBJS=""
for f in ndt2d_kernels ndt2d_match_lane ndt2d_match_small ndt2d_poses_compact ndt2d_build ndt2d_motion ndt2d_scan ndt2d_occupancy ndt2d_device; do
  if [ $f.hip = $FILE ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -I $R/include -I $R/ndt_2d_amd/csrc -c $D/$FILE -o $D/$f.o
    OBJS="$OBJS $D/$f.o"
  else
    OBJS="$OBJS $R/ndt_2d_amd/csrc/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/ndt_2d_amd/csrc/ndt2d_host.o -o $R/experiments/bin/$NAME.so
