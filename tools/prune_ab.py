#!/usr/bin/env python3
"""Dominance-pruned candidate lists (grid.hip, round 5) against the layouts of rounds 2-4, in one process: for each workload one context
per setting of STOCS_GRID_PRUNE (read at context creation), the kernel time of a scoring launch (HIP events), and the scores compared
BITWISE.  usage: python tools/prune_ab.py [Cm C5 dense small ...] [--rounds N] [--div D]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STOCS_PIN_BLAS", "1")
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

args = sys.argv[1:]
rounds = 5
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
div = None
if "--div" in args:
    i = args.index("--div"); div = args[i + 1]; del args[i:i + 2]
names = args or ["Cm", "C5"]
os.environ["STOCS_DEBUG_TIMING"] = "1"          # the grid line on stderr: cell edge, entries per list before / after pruning
out = {}
for name in names:
    m, s, k = synth.workload(name)
    rec = {}
    ref = None
    for prune in ("0", "1"):
        os.environ["STOCS_GRID_PRUNE"] = prune
        if div:
            os.environ["STOCS_GRID_DIV"] = div
        sys.stderr.write("--- %s prune=%s\n" % (name, prune)); sys.stderr.flush()
        est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
        cs, cm = est.get_scene_centroid().astype(np.float64), est.get_model_centroid().astype(np.float64)
        T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
        dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
        est.dev_upload(dT, T)
        for _ in range(3):
            est.score_device(dT, k, dL)        # (the distance field of the patch test fills after 1e9 point queries)
        sc = np.zeros(k, np.float32)
        est.dev_download(dL, sc)
        t = [est.time_score_kernel(dT, k, dL, 10) for _ in range(rounds)]
        rec["prune_" + prune] = {"ms_median": float(np.median(t)), "ms_all": [round(x, 4) for x in t], "Mposes_per_s": k / float(np.median(t)) / 1e3}
        if ref is None:
            ref = sc
        else:
            rec["scores_bitwise_equal"] = bool(np.array_equal(ref.view(np.uint32), sc.view(np.uint32)))
            rec["max_abs_score_diff"] = float(np.abs(ref - sc).max())
        est.dev_free(dT); est.dev_free(dL); est.close()
    rec["speedup"] = rec["prune_0"]["ms_median"] / rec["prune_1"]["ms_median"]
    out[name] = rec
print(json.dumps(out, indent=1))
