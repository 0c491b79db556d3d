import json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth
from model_matching_amd.estimator import StocsEstimator
for nS, nM, K, lattice in [(50000, 12500, 32768, 0.0032), (35000, 8000, 32768, 0.004), (100000, 25000, 16384, 0.0022)]:
    m = synth.make_model(nM); s = synth.make_scene(m, nS, lattice=lattice)
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    cs = est.get_scene_centroid().astype(np.float64); cm = est.get_model_centroid().astype(np.float64)
    T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), K)
    dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(K * 4); est.dev_upload(dT, T)
    out = {}
    for v in (99, 15, 24, 39):
        try:
            est.set_option("lcp_variant", v)
        except Exception as e:
            continue
        est.score_device(dT, K, dL); est.sync()
        o = np.zeros(K, np.float32); est.dev_download(dL, o)
        ms = min(est.time_score_kernel(dT, K, dL, 5) for _ in range(3))
        out[v] = (round(ms, 4), int(np.bitwise_xor.reduce(o.view(np.uint32))))
    print(nS, nM, K, out, flush=True)
    est.close()
