# The wave / item trace of the large search (experiments/README.md, "Where the large search's last 16 % is").
# Here:    bash experiments/build_variant_lib.sh lane_trace ndt2d_match_lane.hip -DNDT2D_LANE_TRACE
# On the GPU box (through gpurun, from the repo root): bash experiments/run_lane_trace.sh
# (split_trace.py / split_check.py need experiments/split_items.patch applied and the library rebuilt)
export NDT2D_HIP_LIB=$PWD/experiments/bin/lane_trace.so
for c in "1.0 0.5" "1.0 0.35" "1.0 0.1"; do timeout 100 python experiments/lane_wave_trace.py $c; done
