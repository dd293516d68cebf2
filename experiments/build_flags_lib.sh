# experiments/bin/<name>.so: the whole library with extra compiler flags for EVERY device translation unit
# (optionally from another source directory: CSRC=/tmp/copy bash experiments/build_flags_lib.sh name flags...)
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
SRC=${CSRC:-$R/ndt_2d_amd/csrc}
mkdir -p $R/experiments/bin/obj_$NAME
OBJS=""
for f in ndt2d_kernels ndt2d_match_lane ndt2d_match_small ndt2d_poses_compact ndt2d_build ndt2d_motion ndt2d_scan ndt2d_occupancy ndt2d_exchange ndt2d_device; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -I $R/include -I $SRC -c $SRC/$f.hip -o $R/experiments/bin/obj_$NAME/$f.o &
  OBJS="$OBJS $R/experiments/bin/obj_$NAME/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/ndt_2d_amd/csrc/ndt2d_host.o -ldl -o $R/experiments/bin/$NAME.so
