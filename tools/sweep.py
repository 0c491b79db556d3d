#!/usr/bin/env python3
"""Config 5 of BASELINE.json: LCP-verify throughput vs problem size (scene / model points), with the
algorithmic-bytes roofline figure of SURVEY.md 8(d).  usage: python tools/sweep.py > profiles/rNN_sweep.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402


def main():
    rows = []
    for nS, nM, K, lattice in [(20000, 5000, 65536, 0.005), (50000, 12500, 32768, 0.0032), (100000, 25000, 16384, 0.0022),
                               (200000, 50000, 16384, 0.0016)]:
        m = synth.make_model(nM)
        s = synth.make_scene(m, nS, lattice=lattice)
        est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
        cs = est.get_scene_centroid().astype(np.float64)
        cm = est.get_model_centroid().astype(np.float64)
        T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), K)
        dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(K * 4)
        est.dev_upload(dT, T)
        est.score_device(dT, K, dL)
        est.sync()
        ms = min(est.time_score_kernel(dT, K, dL, 5) for _ in range(3))
        b_pose = 68 + 52 * nM
        rows.append({"scene_points": nS, "model_points": nM, "candidates": K, "kernel_ms": ms, "poses_per_s": K / (ms * 1e-3),
                     "nn_queries_per_s": K * nM / (ms * 1e-3), "algorithmic_GBps": b_pose * K / (ms * 1e-3) / 1e9,
                     "frac_of_8TBps": b_pose * K / (ms * 1e-3) / 8e12})
        est.dev_free(dT); est.dev_free(dL); est.close()
    print(json.dumps({"sweep": rows}, indent=1))


if __name__ == "__main__":
    main()
