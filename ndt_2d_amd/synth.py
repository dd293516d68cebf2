"""Synthetic workloads of BASELINE.md section 3 / SURVEY.md section 8d.

A closed square room with 0.5 m square pillars on a regular lattice, 720-beam
scans ray-cast by libndt2d_hip.so's host-side generator (ndt2d_synth_scan:
splitmix64 + Box-Muller range noise).  All lattice coordinates and range_max
values are multiples of 0.25 so the NDT extent is binary-exact and the grid
sizes below are reproduced exactly (asserted by the tests).
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import World, dptr

N_BEAMS = 720
NOISE_SIGMA = 0.01

# cfg id -> description.  Search parameters are ScanMatcherNDT's declared
# parameters (reference src/scan_matcher_ndt.cpp:37-44).
CONFIGS = {
    1: dict(name="cfg1-cpu-reference", world=(4.0, 4.0, 0.25), map_lattice=(3, 0.25),
            range_max=4.75, map_seed=1, grid=(41, 41),
            query=dict(true_pose=(0.13, -0.07, 0.031), guess=(0.0, 0.0, 0.0), seed=101),
            search=dict(search_linear_size=0.5, search_linear_resolution=0.05,
                        search_angular_size=0.2, search_angular_resolution=0.01),
            lattice=(21, 40)),
    2: dict(name="cfg2-1gpu-match", world=(4.0, 4.0, 0.25), map_lattice=(3, 0.25),
            range_max=4.75, map_seed=1, grid=(41, 41),
            query=dict(true_pose=(0.13, -0.07, 0.031), guess=(0.0, 0.0, 0.0), seed=101),
            search=dict(search_linear_size=1.0, search_linear_resolution=0.02,
                        search_angular_size=0.5, search_angular_resolution=0.005),
            lattice=(100, 200)),
    3: dict(name="cfg3-1gpu-particles", world=(23.0, 4.0, 0.25), map_lattice=(25, 1.5),
            range_max=7.0, map_seed=3, grid=(201, 201),
            query=dict(true_pose=(1.0, 0.5, 0.3), guess=(1.0, 0.5, 0.3), seed=301),
            particles=dict(n=100000, seed=303)),
    4: dict(name="cfg4-8gpu-loop-closure", world=(4.0, 4.0, 0.25), map_lattice=(3, 0.25),
            range_max=4.75, map_seed=1, grid=(41, 41),
            query=dict(true_pose=(0.13, -0.07, 0.031), guess=(0.0, 0.0, 0.0), seed=101),
            search=dict(search_linear_size=5.0, search_linear_resolution=0.02,
                        search_angular_size=math.pi, search_angular_resolution=0.005),
            lattice=(501, 1257)),
    5: dict(name="cfg5-8gpu-particles", world=(95.0, 5.0, 0.25), map_lattice=(40, 4.5),
            range_max=12.25, map_seed=5, grid=(801, 801),
            query=dict(true_pose=(1.0, 0.5, 0.3), guess=(1.0, 0.5, 0.3), seed=501),
            particles=dict(n=1000000, seed=505)),
}


def world_of(cfg):
    half, pitch, phalf = CONFIGS[cfg]["world"] if isinstance(cfg, int) else cfg
    return World(half, pitch, phalf)


def scan(world, pose, seed, n_beams=N_BEAMS, noise_sigma=NOISE_SIGMA):
    """One scan: robot-frame points[n_beams, 2] seen from `pose` in `world`."""
    L = _capi.lib()
    p = np.ascontiguousarray(pose, dtype=np.float64)
    out = np.zeros((n_beams, 2), dtype=np.float64)
    rc = L.ndt2d_synth_scan(C.byref(world), dptr(p), n_beams, noise_sigma, int(seed), dptr(out))
    if rc != _capi.OK:
        raise _capi.Ndt2dError(rc, "ndt2d_synth_scan")
    return out


def pose_blocked(world, x, y, margin=0.25):
    return bool(_capi.lib().ndt2d_synth_pose_blocked(C.byref(world), x, y, margin))


def uniform(seed, n):
    out = np.zeros(n, dtype=np.float64)
    _capi.lib().ndt2d_synth_uniform(int(seed), n, dptr(out))
    return out


def map_scans(cfg):
    """The scans fed to addScans: a centred k x k pose lattice, theta = 0; poses
    that fall inside / next to a pillar are skipped."""
    c = CONFIGS[cfg]
    w = world_of(cfg)
    k, pitch = c["map_lattice"]
    coords = [(i - (k - 1) / 2.0) * pitch for i in range(k)]
    scans = []
    idx = 0
    for y in coords:
        for x in coords:
            if not pose_blocked(w, x, y):
                scans.append(((x, y, 0.0), scan(w, (x, y, 0.0), c["map_seed"] * 1000003 + idx)))
            idx += 1
    return scans


def query_scan(cfg):
    """(guess pose handed to the matcher as scan->pose, robot-frame points, true pose)."""
    c = CONFIGS[cfg]
    q = c["query"]
    pts = scan(world_of(cfg), q["true_pose"], q["seed"])
    return np.array(q["guess"], dtype=np.float64), pts, np.array(q["true_pose"])


def particles(cfg, n=None):
    """Particles x, y ~ U(room), theta ~ U(-pi, pi)."""
    c = CONFIGS[cfg]
    p = c["particles"]
    n = p["n"] if n is None else n
    half = c["world"][0]
    u = uniform(p["seed"], 3 * n).reshape(n, 3)
    out = np.empty((n, 3), dtype=np.float64)
    out[:, 0] = (2.0 * u[:, 0] - 1.0) * half
    out[:, 1] = (2.0 * u[:, 1] - 1.0) * half
    out[:, 2] = (2.0 * u[:, 2] - 1.0) * math.pi
    return out


def matcher_params(cfg, **override):
    """Keyword arguments for ScanMatcherNDT.initialize for this config."""
    c = CONFIGS[cfg]
    p = dict(ndt_resolution=0.25, laser_max_beams=N_BEAMS)
    p.update(c.get("search", {}))
    p.update(override)
    return dict(range_max=c["range_max"], **p)
