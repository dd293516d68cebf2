# experiments/bin/<name>.so: the library with extra compiler flags for ONE translation unit
#   bash experiments/build_variant_lib.sh <name> <file.hip> <flags...>
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; FILE=$2; shift 2
mkdir -p $R/experiments/bin/obj_$NAME
OBJS=""
for f in ndt2d_kernels ndt2d_match_lane ndt2d_match_small ndt2d_poses_compact ndt2d_build ndt2d_motion ndt2d_scan ndt2d_occupancy ndt2d_exchange ndt2d_device; do
  if [ $f.hip = $FILE ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -I $R/include -I $R/ndt_2d_amd/csrc -c $R/ndt_2d_amd/csrc/$f.hip -o $R/experiments/bin/obj_$NAME/$f.o
    OBJS="$OBJS $R/experiments/bin/obj_$NAME/$f.o"
  else
    OBJS="$OBJS $R/ndt_2d_amd/csrc/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/ndt_2d_amd/csrc/ndt2d_host.o $R/ndt_2d_amd/csrc/ndt2d_build_info.o -ldl -lpthread -o $R/experiments/bin/$NAME.so
