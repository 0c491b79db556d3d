#!/usr/bin/env python3
"""Census (CPU only): length of the candidate list a surviving query scans, by cell edge eps / div.  usage: python tools/list_census.py [Cm|C5] [div]"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth
from scipy.spatial import cKDTree
name=sys.argv[1] if len(sys.argv)>1 else 'Cm'
m,s,k=synth.workload(name)
cs=s.pos.astype(np.float64).mean(0); cm=m.pos.astype(np.float64).mean(0)
T=synth.make_candidates(synth.centred_gt(s.T_gt,cs,cm),k)
sp=s.pos.astype(np.float64)-cs; mp=m.pos.astype(np.float64)-cm
eps=0.005; h=eps/float(sys.argv[2]); r=1.001*eps
o=sp.min(0)-(r+2*h)
tree=cKDTree(sp)
rng=np.random.default_rng(0)
lens=[]
for ci in rng.choice(k,24,replace=False):
    M=T[ci].reshape(4,4).T.astype(np.float64)
    q=mp@M[:3,:3].T+M[:3,3]
    d,_=tree.query(q)
    q=q[d<=1.2*eps][::4]
    cell=np.floor((q-o)/h)
    cen=o+(cell+0.5)*h
    nb=tree.query_ball_point(cen, r+h*0.87)
    for c,ids in zip(cell,nb):
        p=sp[ids]; lo=o+c*h; hi=lo+h
        dd=np.maximum(0,np.maximum(lo-p,p-hi)); 
        lens.append(((dd**2).sum(1)<=r*r).sum())
lens=np.array(lens)
print(name,'survivor list length pct 10/25/50/75/90/99',np.percentile(lens,[10,25,50,75,90,99]),'mean',lens.mean())
nl=(lens+7)//8
print('lines: ',{int(v):float((nl==v).mean()) for v in np.unique(nl)})
