export NDT2D_LANE_SPLIT=0
echo "== in-tree (768 threads, 6 waves/SIMD)"; timeout 200 python experiments/mid_lattice_parts.py 2>&1 | grep -E "items  (3549|6760|13520|23660)"
export NDT2D_HIP_LIB=$PWD/experiments/bin/lane_t1024.so
echo "== 1024 threads, 8 waves/SIMD (scalar exp constants)"; timeout 200 python experiments/mid_lattice_parts.py 2>&1 | grep -E "items  (3549|6760|13520|23660)"
