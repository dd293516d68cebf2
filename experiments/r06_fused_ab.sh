# Cell::addPoint's divisions on the divider against ndt2d_fastdiv.h's divider-free sequence, on the GPU box's host:
# the build phase by phase (DIVIDER=1: the divider) and the mapper's cycle through the C probes (NDT2D_HOST_DIVIDER=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
for r in 1 2 3; do for t in 1 0; do
  echo -n "divider "; DIVIDER=1 experiments/bin/host_build_phases $t
  echo -n "fused   "; experiments/bin/host_build_phases $t
done; done 2>&1 | tee $O/phases.txt
for r in 1 2; do for t in 1 0; do
  echo "divider toy=$t"; NDT2D_HOST_DIVIDER=1 experiments/bin/cycle_breakdown 2000 $t | grep "addScans\|cycle"
  echo "fused   toy=$t"; experiments/bin/cycle_breakdown 2000 $t | grep "addScans\|cycle"
done; done 2>&1 | tee $O/cycle.txt
for r in 1 2; do
  NDT2D_HOST_DIVIDER=1 ndt_2d_amd/ndt2d_latency_probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('divider', d['add_scans_us'], d['mapper_cycle_us'], d['real_lidar_map']['add_scans_us'], d['real_lidar_map']['mapper_cycle_us'])"
  ndt_2d_amd/ndt2d_latency_probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused  ', d['add_scans_us'], d['mapper_cycle_us'], d['real_lidar_map']['add_scans_us'], d['real_lidar_map']['mapper_cycle_us'])"
done 2>&1 | tee $O/probe.txt
