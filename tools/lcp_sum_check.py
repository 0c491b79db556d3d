import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth
from model_matching_amd.estimator import StocsEstimator
from oracle import pyoracle
name = sys.argv[1]
m, s, k = synth.workload(name)
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
orc = pyoracle.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
cs, cm = orc.centroids()
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs.astype(np.float64), cm.astype(np.float64)), k)
got = est.score_transforms(T)
ref, ex = orc.lcp_batch_exact(T, 64)
d1 = np.abs(got - ref); d2 = np.abs(got.astype(np.float64) - ex)
print(name, "vs float oracle: max %.3g, >1e-5: %d;  vs exact sum: max %.3g, >1e-6: %d; oracle float vs exact: max %.3g" % (d1.max(), (d1 > 1e-5).sum(), d2.max(), (d2 > 1e-6).sum(), np.abs(ref - ex).max()))
w = np.argsort(d2)[-3:]
for c in w:
    hg, cg = est.lcp_detail(T[c]); ho, co = orc.lcp_detail(T[c])
    print("  cand", c, "diff", d2[c], "differing hits", int((hg != ho).sum()), "differing counted", int((cg != co).sum()))
