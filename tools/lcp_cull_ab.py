#!/usr/bin/env python3
"""A/B of the patch test of the scoring kernel (stocs_set_option "lcp_cull") and of the model order behind it (median-split
patches, the default, against the Morton order of rounds 1-2: STOCS_MODEL_ORDER), interleaved in ONE process with HIP events on
the contexts' streams; the four score arrays are compared bitwise.  Also times the distance field's fill (first culled call
after a set_scene against the calls that follow).
usage: python tools/lcp_cull_ab.py [Cm|C5|small|dense] [rounds] [candidates]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    m, s, k = synth.workload(name)
    if len(sys.argv) > 3:
        k = int(sys.argv[3])
    ests = {}
    for order in ("kd", "morton"):
        os.environ["STOCS_MODEL_ORDER"] = order
        ests[order] = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    os.environ.pop("STOCS_MODEL_ORDER")
    e0 = ests["kd"]
    cs = e0.get_scene_centroid().astype(np.float64)
    cm = e0.get_model_centroid().astype(np.float64)
    T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
    rep = {"workload": name, "K": k, "nS": e0.nS, "nM": e0.nM}
    res, times, bufs = {}, {}, {}
    for order, est in ests.items():
        dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
        est.dev_upload(dT, T)
        bufs[order] = (dT, dL)
        patches, _, g, _ = est.cull_state(with_field=False)
        rep["patch_radius_mm_%s" % order] = [float(x) for x in np.percentile(patches[:, 3] * 1e3, [10, 50, 90, 100])]
        rep["field_%s" % order] = {"cell_mm": g["g"] * 1e3, "cap_mm": g["cap"] * 1e3, "dims": list(g["dims"])}
        for cull in (0, 2):
            est.set_option("lcp_cull", cull)
            t0 = time.perf_counter()
            est.score_device(dT, k, dL)
            out = np.zeros(k, np.float32)
            est.dev_download(dL, out)
            rep["first_call_ms_%s_cull%d" % (order, cull)] = (time.perf_counter() - t0) * 1e3
            res[(order, cull)] = out
            times[(order, cull)] = []
    for r in range(rounds):
        for (order, cull) in times:
            est = ests[order]
            est.set_option("lcp_cull", cull)
            times[(order, cull)].append(est.time_score_kernel(bufs[order][0], k, bufs[order][1], 10))
    base = res[("kd", 0)]
    for key, t in times.items():
        t = np.array(t)
        rep["%s_cull%d" % key] = {"ms_median": float(np.median(t)), "ms_min": float(t.min()), "Mposes_per_s": k / np.median(t) / 1e3,
                                  "bitwise_equal_to_kd_cull0": bool(np.array_equal(res[key].view(np.uint32), base.view(np.uint32)))}
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
