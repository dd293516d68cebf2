import sys, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/experiments')
import bench_variants as B
r = B.match_case(3, ["auto"], search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.25, search_angular_resolution=0.005)
print(json.dumps(r))
r = B.match_case(5, ["auto"], search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.25, search_angular_resolution=0.005)
print(json.dumps(r))
