#!/usr/bin/env python3
"""Census (CPU only, scipy kd-tree; no library code): which 64-point steps of the sorted model a sphere test against the scene could rule out, for the Morton and the median-split order.  usage: python tools/cull_census.py [Cm|C5] [candidates]"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth
from scipy.spatial import cKDTree
from scipy.ndimage import distance_transform_edt
name = sys.argv[1] if len(sys.argv)>1 else 'Cm'
ncand = int(sys.argv[2]) if len(sys.argv)>2 else 64
m, s, k = synth.workload(name)
cs = s.pos.astype(np.float64).mean(0); cm = m.pos.astype(np.float64).mean(0)
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
sp = s.pos.astype(np.float64)-cs; mp0 = m.pos.astype(np.float64)-cm
tree = cKDTree(sp)
eps=0.005
def morton(mp):
    lo=mp.min(0); ext=(mp.max(0)-lo).max()
    q=np.clip(((mp-lo)/ext*1023).astype(np.int64),0,1023)
    def spread(x):
        x=(x|(x<<16))&0x30000ff; x=(x|(x<<8))&0x300f00f; x=(x|(x<<4))&0x30c30c3; x=(x|(x<<2))&0x9249249; return x
    key=(spread(q[:,2])<<2)|(spread(q[:,1])<<1)|spread(q[:,0])
    return np.argsort(key,kind='stable')
def kdorder(mp, P=64):
    # split so that left gets a multiple of P points (balanced), along the longest axis
    out=[]
    def rec(idx):
        n=len(idx)
        if n<=P: out.append(idx); return
        pts=mp[idx]; ax=np.argmax(pts.max(0)-pts.min(0))
        nl=((n//P+1)//2)*P if n%P else (n//P//2)*P
        if nl==0 or nl>=n: nl=(n//2//P)*P or P
        o=np.argsort(pts[:,ax],kind='stable')
        rec(idx[o[:nl]]); rec(idx[o[nl:]])
    rec(np.arange(len(mp)))
    return np.concatenate(out)
# coarse distance grid
for g_div in (1.0,):
    g=eps/g_div
    org=sp.min(0)-0.08
    dims=np.ceil((sp.max(0)+0.08-org)/g).astype(int)
    occ=np.zeros(dims,bool)
    c=np.floor((sp-org)/g).astype(int); occ[c[:,0],c[:,1],c[:,2]]=True
    dc=distance_transform_edt(~occ)
    Dlb=np.maximum(0,dc-np.sqrt(3))*g
    print('grid',dims,'cells',dims.prod())
for oname,order in (('morton',morton(mp0)),('kd',kdorder(mp0))):
    mp=mp0[order]; M=len(mp); P=64
    npz=(M+P-1)//P
    cen=np.zeros((npz,3)); rad=np.zeros(npz)
    for j in range(npz):
        pts=mp[j*P:(j+1)*P]
        cc=pts.mean(0)
        for it in range(50):  # crude minimal enclosing sphere (Ritter-like iteration)
            d=np.linalg.norm(pts-cc,axis=1); f=pts[np.argmax(d)]; cc=cc+(f-cc)*0.05
        cen[j]=cc; rad[j]=np.linalg.norm(pts-cc,axis=1).max()
    print(name,oname,'radius mm pct 10/50/90/max',np.percentile(rad*1000,[10,50,90,100]))
    rng=np.random.default_rng(0)
    sel=rng.choice(k,ncand,replace=False)
    tot=0; empty=0; cull_exact=0; cull_grid=0
    for ci in sel:
        Mx=T[ci].reshape(4,4).T.astype(np.float64)
        qq=mp@Mx[:3,:3].T+Mx[:3,3]
        d,_=tree.query(qq,k=1)
        surv=d<=1.2*eps
        cc=cen@Mx[:3,:3].T+Mx[:3,3]
        dcen,_=tree.query(cc,k=1)
        pad=np.zeros(npz*P,bool); pad[:M]=surv
        per=pad.reshape(npz,P).sum(1)
        ci3=np.floor((cc-org)/g).astype(int)
        ok=np.all((ci3>=0)&(ci3<dims),axis=1)
        dl=np.zeros(npz); dl[ok]=Dlb[ci3[ok,0],ci3[ok,1],ci3[ok,2]]
        tot+=npz; empty+=(per==0).sum(); cull_exact+=(dcen>rad+1.001*eps).sum(); cull_grid+=(dl>rad+1.001*eps).sum()
    print(' steps',tot,'empty',empty/tot,'cull exact-dist',cull_exact/tot,'cull grid',cull_grid/tot)
