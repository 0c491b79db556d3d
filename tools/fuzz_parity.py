#!/usr/bin/env python3
"""Randomised parity sweep: N small random workloads (model size, scene size, clutter, seeds all drawn), the whole
hot path on the GPU against the CPU oracle: bases + invariants, per-base quad counts, candidate transforms
(bit-exact) and scores (1e-5).  Prints one line per mismatch and a summary.
usage: python tools/fuzz_parity.py [N] [first_seed]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    pyoracle.build()
    bad = 0
    stats = []
    for k in range(n):
        rng = np.random.default_rng(first + k)
        nm = int(rng.integers(120, 420))
        ns = int(rng.integers(500, 2200))
        m = synth.make_model(nm, seed=first + 7 * k)
        s = synth.make_scene(m, ns, seed=first + 11 * k, T_gt=synth.gt_pose(seed=first + 13 * k))
        args = (s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm)
        est = StocsEstimator(*args, build_index=True)
        orc = pyoracle.Oracle(*args, build_index=True)
        seed = int(rng.integers(1, 1 << 30))
        nb = int(rng.integers(20, 80))
        r = orc.run(seed, nb, 200)
        valid, ids, inv = est.sample_bases(seed, nb)
        ok = int(valid.sum()) == r.n_bases
        tot = est.find_congruent_all()
        ok &= tot == r.n_quads_total
        nc = est.make_transforms(200, seed)
        ok &= nc == r.n_candidates
        To, Po, bo = orc.candidates()
        Tg, Pg, lg, bg = est.get_pose_candidates()
        ok &= Tg.shape == To.shape and np.array_equal(To, Tg) and np.array_equal(Po, Pg) and np.array_equal(bo, bg)
        lcp, idx, pose = est.compute_best_transform()
        dl = abs(lcp - r.best_lcp)
        ok &= dl <= 1e-5
        if len(To):
            lo = orc.lcp_batch(To, nthreads=8)
            dmax = float(np.abs(est.get_pose_candidates()[2] - lo).max())
            ok &= dmax <= 1e-5
        else:
            dmax = 0.0
        stats.append((nm, ns, nb, r.n_bases, int(tot), int(nc), dmax))
        if not ok:
            bad += 1
            print("MISMATCH", dict(k=k, nm=nm, ns=ns, seed=seed, nb=nb, bases=(int(valid.sum()), r.n_bases), quads=(int(tot), r.n_quads_total),
                                   cands=(int(nc), r.n_candidates), dl=dl, dmax=dmax), flush=True)
        est.close()
    print(json.dumps({"workloads": n, "mismatches": bad, "total_quads": int(sum(x[4] for x in stats)), "total_candidates": int(sum(x[5] for x in stats)),
                      "max_abs_lcp_diff": max(x[6] for x in stats)}))


if __name__ == "__main__":
    main()
