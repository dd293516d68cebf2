#!/usr/bin/env python3
"""A/B timings of kernel variants and secondary workloads (not the headline bench).
Prints one JSON object; run on an MI355X."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402


def time_match(m, n_th, reps=5):
    ms = []
    for i in range(reps + 1):
        m.match_launch(0, n_th)
        t, _ = m.last_launch_ms()
        if i:
            ms.append(t)
    return float(np.median(ms))


def match_case(cfg, variants, **override):
    m = ScanMatcherNDT(0)
    m.initialize("bench", **synth.matcher_params(cfg, **override))
    m.addScans(synth.map_scans(cfg))
    guess, pts, _ = synth.query_scan(cfg)
    n_th, n_lin, n_b = m.prepare_search(guess, pts)
    units = n_th * n_lin * n_lin * n_b
    out = {"candidates": n_th * n_lin * n_lin, "beams": n_b, "units": units}
    for v in variants:
        m.set_variant(v)
        ms = time_match(m, n_th)
        out[v] = {"kernel_ms": ms, "units_per_s": units / (ms * 1e-3), "variant": m.last_variant()}
    if cfg == 4:
        # most cfg-4 units fall off the 41 x 41 grid (an exact 0 without a
        # cell look-up, in the reference as well): the share that lands inside it, from
        # 4e6 sampled (candidate, beam) pairs
        p = synth.matcher_params(cfg, **override)
        _, gsx, gsy, gcell, gox, goy = m.grid()
        rng = np.random.default_rng(4)
        k = 4_000_000
        th = guess[2] + rng.uniform(-p["search_angular_size"], p["search_angular_size"], k)
        b = pts[rng.integers(0, len(pts), k)]
        lin = p["search_linear_size"]
        x = guess[0] + rng.uniform(-lin, lin, k) + np.cos(th) * b[:, 0] - np.sin(th) * b[:, 1]
        y = guess[1] + rng.uniform(-lin, lin, k) + np.sin(th) * b[:, 0] + np.cos(th) * b[:, 1]
        inside = (x >= gox) & (y >= goy) & (x < gox + gsx * gcell) & (y < goy + gsy * gcell)
        out["in_grid_fraction_sampled"] = float(inside.mean())
        out["in_grid_units_per_s"] = out["auto"]["units_per_s"] * float(inside.mean())
    m.close()
    return out


def default_latency():
    """Per-scan cost of the mapper's local matching step with the plugin defaults
    (reset + addScans(10 scans) + matchScan, reference src/ndt_mapper.cpp:508-515)."""
    scans = synth.map_scans(1) + [synth.map_scans(1)[0]]
    p = synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                             search_angular_size=0.1, search_angular_resolution=0.0025,
                             laser_max_beams=100)
    guess, pts, _ = synth.query_scan(1)
    pose = np.array([0.11, -0.05, 0.02])
    m = ScanMatcherNDT(0)
    m.initialize("local_scan_matcher", **p)
    t_add, t_match = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        m.reset()
        m.addScans(scans)
        t1 = time.perf_counter()
        r = m.matchScan(pose, pts)
        t2 = time.perf_counter()
        t_add.append(t1 - t0)
        t_match.append(t2 - t1)
    kernel_ms = m.last_launch_ms()[0]
    m.close()
    return {"add_scans_ms": float(np.median(t_add)) * 1e3, "match_scan_ms": float(np.median(t_match)) * 1e3,
            "match_kernel_ms": kernel_ms, "candidates": r["n_candidates"], "units": r["n_candidates"] * 100}


def particles_case(cfg, variants):
    import torch
    m = ScanMatcherNDT(0)
    m.initialize("bench", **synth.matcher_params(cfg))
    m.addScans(synth.map_scans(cfg))
    _, pts, _ = synth.query_scan(cfg)
    parts = synth.particles(cfg)
    n_b = m.prepare_beams(pts)
    d_parts = torch.from_numpy(parts).cuda()
    d_scores = torch.zeros(len(parts), dtype=torch.float64, device="cuda")
    d_stats = torch.zeros(8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    units = len(parts) * n_b
    out = {"particles": len(parts), "units": units}
    for v in variants:
        m.set_variant(v)
        ms = []
        for i in range(6):
            m.score_poses_launch(d_parts.data_ptr(), len(parts), d_scores.data_ptr(), d_stats.data_ptr())
            t, _ = m.last_launch_ms()
            if i:
                ms.append(t)
        ms = float(np.median(ms))
        out[v] = {"kernel_ms": ms, "units_per_s": units / (ms * 1e-3), "variant": m.last_variant()}
    if cfg == 4:
        # most cfg-4 units fall off the 41 x 41 grid (an exact 0 without a
        # cell look-up, in the reference as well): the share that lands inside it, from
        # 4e6 sampled (candidate, beam) pairs
        p = synth.matcher_params(cfg, **override)
        _, gsx, gsy, gcell, gox, goy = m.grid()
        rng = np.random.default_rng(4)
        k = 4_000_000
        th = guess[2] + rng.uniform(-p["search_angular_size"], p["search_angular_size"], k)
        b = pts[rng.integers(0, len(pts), k)]
        lin = p["search_linear_size"]
        x = guess[0] + rng.uniform(-lin, lin, k) + np.cos(th) * b[:, 0] - np.sin(th) * b[:, 1]
        y = guess[1] + rng.uniform(-lin, lin, k) + np.sin(th) * b[:, 0] + np.cos(th) * b[:, 1]
        inside = (x >= gox) & (y >= goy) & (x < gox + gsx * gcell) & (y < goy + gsy * gcell)
        out["in_grid_fraction_sampled"] = float(inside.mean())
        out["in_grid_units_per_s"] = out["auto"]["units_per_s"] * float(inside.mean())
    m.close()
    return out


def build_case(cfg):
    """N1: ScanMatcherNDT::addScans, host build + upload vs build on the device."""
    scans = synth.map_scans(cfg)
    params = synth.matcher_params(cfg)
    n_points = sum(len(s[1]) for s in scans)
    out = {"scans": len(scans), "points": n_points}
    for mode in ("host", "device"):
        m = ScanMatcherNDT(0)
        m.initialize("bench", **params)
        m.set_build_mode(mode)
        wall, dev = [], []
        for i in range(6):
            t0 = time.perf_counter()
            m.addScans(scans)
            m.synchronize()
            t1 = time.perf_counter()
            if i:
                wall.append(t1 - t0)
                if mode == "device":
                    dev.append(m.last_launch_ms()[0])
        out[mode] = {"add_scans_ms": float(np.median(wall)) * 1e3}
        if dev:
            out[mode]["device_pipeline_ms"] = float(np.median(dev))
        m.close()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        ref.addScans(scans)
        t.append(time.perf_counter() - t0)
    out["cpu_oracle_ms"] = float(np.median(t)) * 1e3
    return out


if __name__ == "__main__":
    res = {
        "cfg2_match": match_case(2, ["auto", "wave"]),
        "cfg3map_match": match_case(3, ["auto", "wave"], search_linear_size=1.0,
                                    search_linear_resolution=0.02, search_angular_size=0.25,
                                    search_angular_resolution=0.005),
        "cfg4_match_full_lattice": match_case(4, ["auto"]),
        "default_plugin_latency": default_latency(),
        "cfg3_particles": particles_case(3, ["auto", "dense"]),
        "cfg5_particles": particles_case(5, ["auto", "dense"]),
        "ndt_build_cfg1": build_case(1),
        "ndt_build_cfg3": build_case(3),
        "ndt_build_cfg5": build_case(5),
    }
    print(json.dumps(res, indent=1))
