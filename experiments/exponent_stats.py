import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, math
import oracle_lib as O
from ndt_2d_amd import synth, host_build_grid
scans = synth.map_scans(2); p = synth.matcher_params(2)
cells, sx, sy, ox, oy = host_build_grid(0.25, p['range_max'], scans)
occ = (cells[:,5] >= 5)
guess, pts, _ = synth.query_scan(2)
dth = O.search_offsets(0.5, 0.005); dlin = O.search_offsets(1.0, 0.02)
rng = np.random.default_rng(1)
tot=0; act=0; under=0; neg40=0; lanes_occ=0; lanes_sig=0
hist=[]
for _ in range(200):
    ith = rng.integers(len(dth)); ix0 = rng.integers(len(dlin)-8); iy0 = rng.integers(len(dlin)-8)
    c,s = math.cos(dth[ith]), math.sin(dth[ith])
    oxp = pts[:,0]*c - pts[:,1]*s; oyp = pts[:,0]*s + pts[:,1]*c
    PX = (oxp[:,None,None] + dlin[ix0:ix0+8][None,:,None] + 0*dlin[None,None,iy0:iy0+8]).reshape(len(pts),64)
    PY = (oyp[:,None,None] + 0*dlin[ix0:ix0+8][None,:,None] + dlin[None,None,iy0:iy0+8]).reshape(len(pts),64)
    gx = np.floor((PX-ox)/0.25).astype(int); gy = np.floor((PY-oy)/0.25).astype(int)
    inside = (gx>=0)&(gx<sx)&(gy>=0)&(gy<sy)
    idx = np.where(inside, gy*sx+gx, 0)
    o = inside & occ[idx]
    rec = cells[idx]
    q0 = PX-rec[...,0]; q1 = PY-rec[...,1]
    e = -0.5*(q0*(rec[...,2]*q0+rec[...,3]*q1) + q1*(rec[...,3]*q0+rec[...,4]*q1))
    e = np.where(o, e, -np.inf)
    a = o.any(axis=1)
    tot += len(pts); act += a.sum()
    emax = e.max(axis=1)
    under += (a & (emax < -745.2)).sum()
    neg40 += (a & (emax < -40)).sum()
    lanes_occ += o.sum(); lanes_sig += (e > -40).sum()
print('active iter frac', act/tot, 'of those all-underflow', under/act, ' all < -40:', neg40/act)
print('occupied lane-units', lanes_occ/(tot*64), ' significant (e>-40)', lanes_sig/(tot*64))
