import sys, math
sys.path.insert(0,'/root/repo')
import numpy as np
from ndt_2d_amd import host_build_grid, synth
cfg=int(sys.argv[1]) if len(sys.argv)>1 else 3
scans=synth.map_scans(cfg); p=synth.matcher_params(cfg)
cells,sx,sy,ox,oy=host_build_grid(0.25,p["range_max"],scans)
occ=(cells[:,5]>=5).reshape(sy,sx)
_,pts,_=synth.query_scan(cfg)
parts=synth.particles(cfg, 20000)
print("grid",sx,sy,"occupied fraction",occ.mean())
# chebyshev distance map up to R
R=12
D=np.full(occ.shape, R+1, dtype=np.int32)
cur=occ.copy()
D[occ]=0
for d in range(1,R+1):
    nxt=cur.copy()
    nxt[1:,:]|=cur[:-1,:]; nxt[:-1,:]|=cur[1:,:]
    cur=nxt.copy()
    nxt[:,1:]|=cur[:,:-1]; nxt[:,:-1]|=cur[:,1:]
    # 8-neighbourhood: the two passes above give chebyshev dilation by 1
    new=nxt & (D>R)
    D[new]=d
    cur=nxt
for RUN in (8,12,16,24):
    n=len(pts)
    chunks=8; clen=(n+chunks-1)//chunks
    runs=[]
    for c in range(chunks):
        k0=c*clen; k1=min(n,k0+clen)
        b=k0
        while b<k1:
            e=min(k1,b+RUN); q=pts[b:e]
            ctr=(q.min(axis=0)+q.max(axis=0))/2
            r=np.max(np.hypot(q[:,0]-ctr[0],q[:,1]-ctr[1]))
            runs.append((b,e,ctr,r)); b=e
    rr=np.array([r for _,_,_,r in runs])
    tot=0; maybe=0; beams_maybe=0
    c,s=np.cos(parts[:,2]),np.sin(parts[:,2])
    for (b,e,ctr,r) in runs:
        X=parts[:,0]+c*ctr[0]-s*ctr[1]; Y=parts[:,1]+s*ctr[0]+c*ctr[1]
        gx=np.floor((X-ox)/0.25).astype(int); gy=np.floor((Y-oy)/0.25).astype(int)
        need=math.ceil(r/0.25+0.01)+1
        inside=(gx>=0)&(gx<sx)&(gy>=0)&(gy<sy)
        # outside: distance to grid box
        dx=np.maximum(0,np.maximum(-gx,gx-(sx-1))); dy=np.maximum(0,np.maximum(-gy,gy-(sy-1)))
        dout=np.maximum(dx,dy)
        d=np.where(inside, D[np.clip(gy,0,sy-1),np.clip(gx,0,sx-1)], np.maximum(dout, 0)+0)
        # outside the grid: nearest occupied is at least dout away; conservative: treat as D at clamped cell + dout
        d=np.where(inside,d,D[np.clip(gy,0,sy-1),np.clip(gx,0,sx-1)]+dout)
        m=(d<=need) if need<=R else np.ones(len(parts),bool)
        tot+=len(parts); maybe+=m.sum(); beams_maybe+=m.sum()*(e-b)
    print("run %2d: %3d runs, radius median %.2f m max %.2f m: (particle,run) pairs that stay %.3f ; beams still screened %.3f"%(RUN,len(runs),np.median(rr),rr.max(),maybe/tot,beams_maybe/(len(parts)*n)))
