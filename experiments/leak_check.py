import sys
sys.path.insert(0, "/root/repo")
import torch
from ndt_2d_amd import ScanMatcherNDT, synth
scans = synth.map_scans(1); guess, pts, _ = synth.query_scan(1)
def once(ids, ex):
    m = ScanMatcherNDT(device_ids=ids); m.set_exchange(ex); m.set_multi_min_units(0)
    m.initialize("x", **synth.matcher_params(1)); m.addScans(scans); m.matchScan(guess, pts); m.close()
once([0,0],"host"); once([0],"rccl")
torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]
for i in range(30): once([0,0,0],"host")
torch.cuda.synchronize(); f1 = torch.cuda.mem_get_info()[0]
for i in range(10): once([0],"rccl")
torch.cuda.synchronize(); f2 = torch.cuda.mem_get_info()[0]
print("host-exchange matchers: %.1f MB per create/destroy; rccl: %.1f MB" % ((f0-f1)/30/1e6, (f1-f2)/10/1e6))
