# experiments/bin/trace.so: the library with the small-lattice search's trace points compiled in
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/experiments/bin/trace_obj
for f in ndt2d_kernels ndt2d_match_lane ndt2d_match_small ndt2d_poses_compact ndt2d_build ndt2d_motion ndt2d_scan ndt2d_occupancy ndt2d_exchange ndt2d_device; do
  # (every unit every time: a header change -- a struct that gained a field -- otherwise leaves
  # stale objects behind, and a library of mixed layouts runs, wrongly, and measures nonsense)
  if true; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DNDT2D_SMALL_TRACE -I $R/include -I $R/ndt_2d_amd/csrc -c $R/ndt_2d_amd/csrc/$f.hip -o $R/experiments/bin/trace_obj/$f.o &
  fi
done
wait
g++ -O3 -std=c++17 -ffp-contract=off -fPIC -I $R/include -I $R/ndt_2d_amd/csrc -c $R/ndt_2d_amd/csrc/ndt2d_host.cpp -o $R/experiments/bin/trace_obj/ndt2d_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/experiments/bin/trace_obj/*.o $R/ndt_2d_amd/csrc/ndt2d_build_info.o -ldl -lpthread -o $R/experiments/bin/trace.so
