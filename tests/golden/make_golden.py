#!/usr/bin/env python3
"""Generates tests/golden/tiny_golden.npz with the CPU oracle (oracle/stocs_oracle.cpp) on the seeded
"tiny" synthetic workload.  The reference holds no golden vectors and cannot be
built or run here, so these vectors pin the ORACLE over time (regression), and the HIP path against
them on the GPU box; they are not outputs of the reference.   Run:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from model_matching_amd import synth
    from oracle import pyoracle as po
    m, s, k = synth.workload("tiny")
    o = po.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm)
    cs, cm = o.centroids()
    nrm = po.normalize_rows(s.nrm)
    rng = np.random.default_rng(2024)
    pair_idx = rng.integers(0, len(s.pos), (512, 2)).astype(np.int32)
    ppf = np.array([po.ppf_compute(s.pos[i], nrm[i], s.pos[j], nrm[j]) for i, j in pair_idx], np.int32)
    seed = 20241004
    bases_valid, bases_ids, bases_inv = [], [], []
    for a in range(32):
        ok, ids, inv = o.sample_class_base(seed, a)
        bases_valid.append(ok); bases_ids.append(ids.copy()); bases_inv.append(inv.copy())
    bases_valid = np.array(bases_valid); bases_ids = np.array(bases_ids, np.int32); bases_inv = np.array(bases_inv, np.float32)
    quad_counts, quad_first = [], []
    for a in range(32):
        if bases_valid[a]:
            q = o.find_congruent(bases_ids[a], float(bases_inv[a][0]), float(bases_inv[a][1]))
            quad_counts.append(len(q))
            quad_first.append(q[:8].reshape(-1).tolist() + [-1] * (32 - 4 * min(len(q), 8)))
        else:
            quad_counts.append(-1); quad_first.append([-1] * 32)
    r = o.run(seed, 100, 200)
    T, P, b = o.candidates()
    lcp = o.lcp_batch(T)
    Tc = synth.make_candidates(synth.centred_gt(s.T_gt, cs.astype(np.float64), cm.astype(np.float64)), 128, seed=77)
    lcp_c = o.lcp_batch(Tc)
    hit, counted = o.lcp_detail(Tc[int(np.argmax(lcp_c))])
    out = os.path.join(ROOT, "tests", "golden", "tiny_golden.npz")
    np.savez_compressed(out, seed=seed, centroid_scene=cs, centroid_model=cm, pair_idx=pair_idx, ppf=ppf,
                        bases_valid=bases_valid, bases_ids=bases_ids, bases_inv=bases_inv,
                        quad_counts=np.array(quad_counts, np.int64), quad_first=np.array(quad_first, np.int32),
                        run_counts=np.array([r.n_bases, r.n_quads_total, r.n_candidates, r.best_index], np.int64),
                        run_best_lcp=np.float32(r.best_lcp), run_best_pose=np.array(r.best_pose16, np.float32),
                        cand_T=T, cand_base=b, cand_lcp=lcp, synth_T=Tc, synth_lcp=lcp_c,
                        best_hit=hit, best_counted=counted)
    print("wrote", out, os.path.getsize(out), "bytes")
    # summary of oracle runs on the fixtures derived from the reference's example data
    import json
    summ = {}
    for name in ("ycb_024_bowl", "linemod_obj_06", "packed_dove"):
        d = np.load(os.path.join(ROOT, "tests", "golden", "example_%s.npz" % name))
        o = po.Oracle(d["scene_pos"], d["scene_nrm"], d["scene_prob"], d["scene_pixel"], d["model_pos"], d["model_nrm"])
        r = o.run(7, 100, 200)
        summ[name] = dict(nS=int(len(d["scene_pos"])), nM=int(len(d["model_pos"])), seed=7, n_bases=r.n_bases, n_quads=r.n_quads_total,
                          n_candidates=r.n_candidates, best_index=r.best_index, best_lcp=float(np.float32(r.best_lcp)),
                          best_pose16=[float(np.float32(v)) for v in r.best_pose16])
    json.dump(summ, open(os.path.join(ROOT, "tests", "golden", "example_summary.json"), "w"), indent=1)
    print("wrote example_summary.json")


if __name__ == "__main__":
    main()
