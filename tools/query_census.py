#!/usr/bin/env python3
"""Census of the nearest-neighbour queries of a workload's candidate batch (CPU only, scipy kd-tree; no library code involved):
how many queries have a scene point within epsilon, how many such points, how far the nearest is, and how many points the 27
epsilon-cells around a query hold -- the numbers behind the choice of the list layout for dense scenes.
usage: python tools/query_census.py [C5|Cm|dense|small]"""
import numpy as np, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth
from scipy.spatial import cKDTree
name = sys.argv[1] if len(sys.argv)>1 else 'C5'
m, s, k = synth.workload(name)
cs = s.pos.astype(np.float64).mean(0); cm = m.pos.astype(np.float64).mean(0)
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
sp = s.pos.astype(np.float64)-cs; mp = m.pos.astype(np.float64)-cm
tree = cKDTree(sp)
eps=0.005
rng=np.random.default_rng(0)
sel = rng.choice(k, 48, replace=False)
tot=0; hit=0; nn_d=[]; cnt_eps=[]; cnt_box3=[]
h=eps
org = sp.min(0)-2*eps
cell = np.floor((sp-org)/h).astype(np.int64)
dims = cell.max(0)+2
lin = (cell[:,2]*dims[1]+cell[:,1])*dims[0]+cell[:,0]
occ = np.bincount(lin, minlength=int(dims.prod()))
print('scene pts',len(sp),'occupied eps-cells',(occ>0).sum(),'pts per occupied cell',len(sp)/(occ>0).sum(), 'grid dims',dims, 'max per cell', occ.max())
occ3 = occ.reshape(dims[2],dims[1],dims[0])
# 27-cell sums
from scipy.ndimage import uniform_filter
box27 = np.zeros_like(occ3)
pad = np.pad(occ3,1)
for dz in range(3):
  for dy in range(3):
    for dx in range(3):
      box27 += pad[dz:dz+dims[2],dy:dy+dims[1],dx:dx+dims[0]]
for ci in sel:
    M = T[ci].reshape(4,4).T.astype(np.float64)
    q = mp@M[:3,:3].T+M[:3,3]
    d,i = tree.query(q, k=1, distance_upper_bound=eps)
    hmask = np.isfinite(d)
    tot+=len(q); hit+=hmask.sum(); nn_d.append(d[hmask])
    c = tree.query_ball_point(q[::50], eps, return_length=True)
    cnt_eps.append(c)
    qc = np.floor((q-org)/h).astype(np.int64)
    ok = np.all((qc>=0)&(qc<dims),axis=1)
    b = np.zeros(len(q),np.int64)
    b[ok] = box27[qc[ok,2],qc[ok,1],qc[ok,0]]
    cnt_box3.append(b)
nn_d=np.concatenate(nn_d); cnt_eps=np.concatenate(cnt_eps); cnt_box3=np.concatenate(cnt_box3)
print('queries',tot,'hit frac',hit/tot)
print('NN dist (mm) pct 10/50/90/99', np.percentile(nn_d*1000,[10,50,90,99]))
print('pts within eps: mean over all q',cnt_eps.mean(),'mean over hits',cnt_eps[cnt_eps>0].mean())
print('27-cell entries: mean over all',cnt_box3.mean(),'frac nonzero',(cnt_box3>0).mean(),'mean over nonzero',cnt_box3[cnt_box3>0].mean())
