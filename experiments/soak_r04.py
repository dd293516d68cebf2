"""Round-4 soak: the multi-device matcher (three contexts, host exchange; one context, RCCL
exchange), the near-tie adjudication and the host single-pose path driven for `seconds` each in
one process; every repetition must return identical bits, and nothing may hang or leak
(device memory is read before and after)."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
scans = synth.map_scans(1)
guess, pts, _ = synth.query_scan(1)
parts = synth.particles(3, 50000); parts[:, :2] *= 4.0 / 23.0
free0 = torch.cuda.mem_get_info()[0]
report = {}
for label, ids, exchange in (("3ctx-host", [0, 0, 0], "host"), ("1ctx-rccl", [0], "rccl")):
    m = ScanMatcherNDT(device_ids=ids); m.set_exchange(exchange); m.set_multi_min_units(0)
    m.initialize("soak", **synth.matcher_params(2))
    first = None; n = 0; t0 = time.time()
    while time.time() - t0 < seconds:
        m.reset(); m.addScans(scans)
        r = m.matchScan(guess, pts)
        w, mean, cov = pf_measure(m, parts, pts)
        s = m.scorePoints(pts[::8], guess)
        blob = (r["score"], r["best_index"], r["pose"].tobytes(), r["covariance"].tobytes(), w.tobytes(), mean.tobytes(), s)
        if first is None: first = blob
        assert blob == first, (label, n)
        n += 1
    assert m.matcher_variant().startswith("multi[")
    report[label] = n
    m.close()
# adjudication: a tie lattice, marked every time, settled every time
CELL = np.array([[2.0, 2.0], [3.0, 2.0], [1.0, 2.0], [2.0, 3.0], [2.0, 1.0], [2.5, 2.5], [1.5, 1.5], [2.5, 1.5], [1.5, 2.5]])
P = dict(ndt_resolution=4.0, range_max=8.0, laser_max_beams=100, search_linear_size=0.1875, search_linear_resolution=0.125,
         search_angular_size=0.001, search_angular_resolution=0.002)
m = ScanMatcherNDT(0); m.initialize("ties", **P); m.addScans([((0.0, 0.0, 0.0), CELL)])
rng = np.random.default_rng(5); d = np.round(rng.uniform(-0.4, 0.4, (12, 2)) * 4096) / 4096
beams = np.concatenate([2.0 + d, 2.0 - d]); first = None; n = 0; t0 = time.time()
while time.time() - t0 < seconds / 2:
    r = m.matchScan((0.0, 0.0, 0.001), beams)
    blob = (r["score"], r["best_index"], r["pose"].tobytes())
    if first is None: first = blob
    assert blob == first; n += 1
assert m.adjudication_stats()[0] == n
report["adjudicated"] = n
m.close()
torch.cuda.synchronize()
report["device_memory_delta_MB"] = (free0 - torch.cuda.mem_get_info()[0]) / 1e6
print(report)
