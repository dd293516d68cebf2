"""Randomised check of the host NDT build (csrc/ndt2d_host.cpp HostNdt::add_scan, round 6: a scan's four
quarters side by side): scan sets of 1-11 scans x 1-1440 beams, uniform / constant / room-shaped /
three-valued ranges (degenerate cells with NaN information included), cell sizes 0.05-1 m, against the
sequential loop (NDT2D_BUILD_SEQUENTIAL) and the CPU oracle, compared BIT FOR BIT (uint64 views: NaN
cells count).  CPU only:   python experiments/fuzz_host_build.py [seed]   (100 s per run; round 6:
4 seeds, 142,850 cases, no difference)"""
import sys, math, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from ndt_2d_amd import host_build_grid
from ndt_2d_amd.scan_matcher import BUILD_SEQUENTIAL
import oracle_lib as O
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
t0=time.time(); n=0; bad=0
while time.time()-t0 < 100:
    n_scans=int(rng.integers(1,12)); nb=int(rng.choice([1,3,7,31,32,33,64,100,255,360,719,720,721,1440]))
    res=float(rng.choice([0.05,0.1,0.25,0.3,0.5,1.0])); rmax=float(rng.choice([0.5,2.0,4.75,12.0]))
    spread=float(rng.choice([0.0,0.01,0.3,2.0])); reach=float(rng.choice([0.02,0.2,1.0,5.0,30.0]))
    scans=[]
    for _ in range(n_scans):
        pose=(float(rng.uniform(-spread,spread)),float(rng.uniform(-spread,spread)),float(rng.uniform(-math.pi,math.pi)))
        ang=np.linspace(-math.pi,math.pi,nb,endpoint=False)
        mode=rng.integers(0,4)
        if mode==0: r=rng.uniform(0,reach,nb)
        elif mode==1: r=np.full(nb,reach)*rng.uniform(0.99,1.01,nb)
        elif mode==2: r=np.minimum(reach/np.maximum(np.abs(np.cos(ang)),1e-3),reach/np.maximum(np.abs(np.sin(ang)),1e-3))+rng.normal(0,0.01,nb)
        else: r=rng.choice([0.0,reach,reach*0.5],nb)
        scans.append((pose,np.stack([r*np.cos(ang),r*np.sin(ang)],axis=1)))
    try:
        a=host_build_grid(res,rmax,scans); b=host_build_grid(res,rmax,scans,BUILD_SEQUENTIAL)
    except Exception as e:
        print('ERR',e); bad+=1; continue
    m=O.ScanMatcherNDT(); m.initialize(ndt_resolution=res, range_max=rmax); m.addScans(scans)
    c6=np.ascontiguousarray(m.ndt.cells6()); ok = a[1:]==b[1:] and np.array_equal(a[0].view(np.uint64),b[0].view(np.uint64)) and np.array_equal(a[0].view(np.uint64), c6.view(np.uint64))
    if not ok:
        bad+=1; print('MISMATCH', n_scans, nb, res, rmax, spread, reach)
    n+=1
print('cases',n,'bad',bad)
