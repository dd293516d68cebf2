export NDT2D_HIP_LIB=$PWD/experiments/bin/lane_trace.so
for f in 0 1024; do echo "== FINE=$f"; NDT2D_LANE_FINE_ITEMS=$f timeout 100 python experiments/lane_wave_trace.py 1.0 0.1 2>&1 | grep -E "^lin|wave end|last item start|items per wave|longest item|mean wave life|item duration|the ten items" | cut -c1-230; done
