# round 3, first A/B session: particle kernel (r02 build / statistics sums in LDS / five waves
# per SIMD) and the large-map search with four vs six waves per SIMD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab1
mkdir -p $O
for L in poses_r02 "" poses_w5; do
  for i in 1 2; do
    echo "== lib ${L:-current}" >> $O/particles.txt
    if [ -n "$L" ]; then NDT2D_HIP_LIB=$PWD/experiments/bin/$L.so python experiments/particles_ab.py >> $O/particles.txt 2>&1
    else python experiments/particles_ab.py >> $O/particles.txt 2>&1; fi
  done
done
for g in 0 1 0 1; do
  echo "== NDT2D_LANE_GATHER6=$g" >> $O/bigmap.txt
  NDT2D_LANE_GATHER6=$g python experiments/big_map_search.py >> $O/bigmap.txt 2>&1
done
