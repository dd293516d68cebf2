for lib in in-tree build_old in-tree build_old; do
  if [ "$lib" != "in-tree" ]; then export NDT2D_HIP_LIB=$PWD/experiments/bin/$lib.so; else unset NDT2D_HIP_LIB; fi
  echo "== lib $lib"; timeout 300 python experiments/build_ab.py 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
for lib in in-tree build_old; do
  if [ "$lib" != "in-tree" ]; then export NDT2D_HIP_LIB=$GRAFT_REPO_ROOT/experiments/bin/$lib.so; else unset NDT2D_HIP_LIB; fi
  rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r05b/build_$lib -o t -- python3 $GRAFT_REPO_ROOT/experiments/build_ab.py > /dev/null 2>&1
  echo "== kernels $lib"; python3 $GRAFT_REPO_ROOT/experiments/rocpd_kernels.py $GRAFT_REPO_ROOT/gpurun_out/r05b/build_$lib/t_results.db cells_kernel cell_sums_kernel points_kernel segments
done
