import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, math
import oracle_lib as O
from ndt_2d_amd import synth, host_build_grid
scans = synth.map_scans(2); p = synth.matcher_params(2)
cells, sx, sy, ox, oy = host_build_grid(0.25, p['range_max'], scans)
occ = (cells[:,5] >= 5).reshape(sy, sx)
guess, pts, _ = synth.query_scan(2)
dth = O.search_offsets(0.5, 0.005); dlin = O.search_offsets(1.0, 0.02)
rng = np.random.default_rng(0)
def cell_occ(px, py):
    gx = np.floor((px-ox)/0.25).astype(int); gy = np.floor((py-oy)/0.25).astype(int)
    inside = (gx>=0)&(gx<sx)&(gy>=0)&(gy<sy)
    o = np.zeros(px.shape, bool)
    o[inside] = occ[gy[inside], gx[inside]]
    return o
# v1: wave = candidate, lanes = 64 consecutive beams
tot=0; act_units=0; act_waves=0; nw=0
for _ in range(300):
    ith = rng.integers(len(dth)); ix = rng.integers(len(dlin)); iy = rng.integers(len(dlin))
    c,s = math.cos(dth[ith]), math.sin(dth[ith])
    px = pts[:,0]*c - pts[:,1]*s + dlin[ix]; py = pts[:,0]*s + pts[:,1]*c + dlin[iy]
    o = cell_occ(px,py)
    act_units += o.sum(); tot += len(o)
    # waves: beams lane+64j -> wave-iteration j covers beams [64j, 64j+64)
    for j in range(12):
        seg = o[64*j:64*j+64]
        if len(seg): nw+=1; act_waves += seg.any()
print('v1: active unit fraction', act_units/tot, ' active wave-iteration fraction', act_waves/nw)
# v2: wave = 8x8 patch of (ix,iy), loop beams
for P in (8, 4, 16):
    act=0; n=0
    for _ in range(60):
        ith = rng.integers(len(dth)); ix0 = rng.integers(len(dlin)-P); iy0 = rng.integers(len(dlin)-P)
        c,s = math.cos(dth[ith]), math.sin(dth[ith])
        oxp = pts[:,0]*c - pts[:,1]*s; oyp = pts[:,0]*s + pts[:,1]*c
        dx = dlin[ix0:ix0+P]; dy = dlin[iy0:iy0+P]
        PX = oxp[:,None,None] + dx[None,:,None] + 0*dy[None,None,:]
        PY = oyp[:,None,None] + 0*dx[None,:,None] + dy[None,None,:]
        o = cell_occ(PX.reshape(len(pts),-1), PY.reshape(len(pts),-1))
        act += o.any(axis=1).sum(); n += len(pts)
    print('v2 patch %dx%d: active wave-iteration fraction %.3f' % (P,P,act/n))
# v2 with 64x1 strips
act=0;n=0
for _ in range(60):
    ith = rng.integers(len(dth)); ix0 = rng.integers(len(dlin)); iy0 = rng.integers(len(dlin)-64)
    c,s = math.cos(dth[ith]), math.sin(dth[ith])
    oxp = pts[:,0]*c - pts[:,1]*s; oyp = pts[:,0]*s + pts[:,1]*c
    PX = oxp[:,None] + dlin[ix0] + 0*dlin[None,iy0:iy0+64]; PY = oyp[:,None] + dlin[None,iy0:iy0+64]
    o = cell_occ(PX,PY); act += o.any(axis=1).sum(); n+=len(pts)
print('v2 strip 1x64: active wave-iteration fraction %.3f' % (act/n))
