#!/usr/bin/env python3
"""A model beyond 16 384 points (15-bit ids: the packed quad key holds the ids alone) through phases 1-4 on the GPU, quads of
the first bases against the oracle (the checker; its all-pairs index of 18 000 points takes a while on one core).
usage: python tools/big_model_check.py [n_model] [n_bases_checked]"""
import json
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402


def main():
    nM = int(sys.argv[1]) if len(sys.argv) > 1 else 18000
    ncheck = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    m = synth.make_model(nM, seed=4711)
    s = synth.make_scene(m, 20000, seed=4712)
    t = time.perf_counter()
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=True)
    est.sync()
    rep = {"model_points": int(est.nM), "scene_points": int(est.nS), "ctx_create_incl_index_s": time.perf_counter() - t, "index": est.index_stats()}
    valid, ids, inv = est.sample_bases(1234, 24)
    t = time.perf_counter(); nq = est.find_congruent_all(); rep["congruent_ms"] = (time.perf_counter() - t) * 1e3
    nc = est.make_transforms(200, 1234)
    lcp, idx, pose = est.compute_best_transform()
    P = pose.reshape(4, 4).T
    dR = P[:3, :3].T @ s.T_gt[:3, :3]
    rep.update(bases=int(valid.sum()), quads=int(nq), candidates=int(nc), best_lcp=float(lcp),
               rot_err_deg=float(np.degrees(np.arccos(min(1.0, (np.trace(dR) - 1) / 2)))), tr_err_mm=float(np.linalg.norm(P[:3, 3] - s.T_gt[:3, 3]) * 1e3))
    sizes = [int(est.L.stocs_quad_count(est.h, k)) if hasattr(est.L, "stocs_quad_count") else len(est.get_quads(k)) for k in range(int(valid.sum()))]
    rep["quads_per_base_first_8"] = sizes[:8]
    if ncheck:
        from oracle import pyoracle
        t = time.perf_counter()
        orc = pyoracle.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm)
        rep["oracle_create_s"] = time.perf_counter() - t
        checked, slot = [], 0
        for a in range(24):
            if not valid[a]:
                continue
            if len(checked) < ncheck and 0 < sizes[slot] < 2_000_000:
                qo = orc.find_congruent(ids[a], float(inv[a][0]), float(inv[a][1]))
                qg = est.get_quads(slot)
                checked.append({"slot": slot, "quads": int(len(qo)), "equal": bool(np.array_equal(qo, qg)), "max_id": int(qg.max()) if len(qg) else -1})
            slot += 1
        rep["oracle_checks"] = checked
        rep["all_equal"] = all(c["equal"] for c in checked) and len(checked) > 0
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
