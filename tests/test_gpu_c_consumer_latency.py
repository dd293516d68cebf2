"""The plain-C latency probe (ndt_2d_amd/tools/latency_probe.c, the C-ABI as a C host
calls it) runs the plugin-default search; its winner must be the Python path's."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_probe_matches_the_python_path():
    from ndt_2d_amd import ScanMatcherNDT, build, synth
    build.build_all()
    r = subprocess.run([build.PROBE_PATH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert "small-lattice" in out["variant"]
    m = ScanMatcherNDT(0)
    m.initialize("probe", **synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                                                 search_angular_size=0.1, search_angular_resolution=0.0025,
                                                 laser_max_beams=100))
    m.addScans(synth.map_scans(1))
    _, pts, _ = synth.query_scan(1)
    want = m.matchScan((0.11, -0.05, 0.02), pts)
    assert np.array_equal(out["check_pose"], want["pose"])
    assert out["check_score"] == want["score"]
    for key in ("match_scan_us", "score_scan_us", "add_scans_us", "mapper_cycle_us"):
        assert 0.05 < out[key] < 5000.0    # (scoreScan of 100 beams is scored on the host: ~1 us)
    # the unchanged ParticleFilter::measure loop (500 scorePoints calls) through the host path
    assert out["measure_500_particles_unchanged_loop_us"] < 2000.0
