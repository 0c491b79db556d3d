"""Differential tests of the whole pipeline against the oracle under NON-default parameters and at the
full size of the metric configuration (phases 1-3; the LCP kernel is covered at full size in
test_lcp_gpu.py).  Integer / index outputs bit-exact, transforms bit-exact, scores within 1e-5."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LCP_TOL = 1e-5


def _pair(workload, **prm):
    from model_matching_amd import capi, synth
    from model_matching_amd.estimator import StocsEstimator
    from oracle import pyoracle
    m, s, k = synth.workload(workload)
    gp = capi.default_params(**prm)
    op = pyoracle.default_params(**{k_: v for k_, v in prm.items() if k_ != "lcp_normal_angle"})
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, params=gp, build_index=True)
    orc = pyoracle.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, params=op)
    return m, s, est, orc


@pytest.mark.parametrize("prm", [
    dict(distance_threshold=0.008, ppf_tr_discretization=10, ppf_rot_discretization=10),
    dict(distance_threshold=0.003, ppf_tr_discretization=4, ppf_rot_discretization=6, plane_threshold=0.03, min_distance_base=0.02),
    dict(internal_angle_threshold=20.0, ppf_tr_discretization=5, ppf_rot_discretization=15),
])
def test_pipeline_with_other_parameters(prm):
    from model_matching_amd import synth
    m, s, est, orc = _pair("tiny", **prm)
    seed = 31
    valid, ids, inv = est.sample_bases(seed, 40)
    n_ok = 0
    for a in range(40):
        ok, oi, ov = orc.sample_class_base(seed, a)
        assert ok == bool(valid[a]), (prm, a)
        if ok:
            assert np.array_equal(oi, ids[a]) and np.array_equal(ov, inv[a]), (prm, a)
            n_ok += 1
    total = est.find_congruent_all()
    slot, tot_o = 0, 0
    for a in range(40):
        if not valid[a]:
            continue
        qo = orc.find_congruent(ids[a], float(inv[a][0]), float(inv[a][1]))
        qg = est.get_quads(slot); slot += 1
        assert np.array_equal(qo, qg), (prm, a, qo.shape, qg.shape)
        tot_o += len(qo)
        for q in qo[:3]:
            oko, To, Po = orc.rigid_transform(ids[a], q)
            okg, Tg, Pg = est.get_rigid_transform_from_congruent_pair(ids[a], q)
            assert oko == okg and (not oko or (np.array_equal(To, Tg) and np.array_equal(Po, Pg)))
    assert total == tot_o
    cs, cm = orc.centroids()
    T = synth.make_candidates(synth.centred_gt(s.T_gt, cs.astype(np.float64), cm.astype(np.float64)), 300)
    assert np.abs(est.score_transforms(T) - orc.lcp_batch(T)).max() <= LCP_TOL
    hg, cg = est.lcp_detail(T[0]); ho, co = orc.lcp_detail(T[0])
    assert np.array_equal(hg, ho) and np.array_equal(cg, co)


def test_phases_1_to_3_at_metric_size():
    """Cm: 20 000-point scene, 5 000-point model (25 M indexed pairs): sampling, index queries, congruent
    sets of two bases and their transforms against the oracle."""
    m, s, est, orc = _pair("Cm")
    n_pairs, n_buckets, n_keys = est.index_stats()
    assert n_pairs == 5000 * 4999
    seed = 1234
    valid, ids, inv = est.sample_bases(seed, 16)
    for a in range(16):
        ok, oi, ov = orc.sample_class_base(seed, a)
        assert ok == bool(valid[a]), a
        if ok:
            assert np.array_equal(oi, ids[a]) and np.array_equal(ov, inv[a]), a
    assert valid.sum() >= 10
    rng = np.random.default_rng(0)
    nrm = None
    for a in np.nonzero(valid)[0][:3]:
        # the two index look-ups of this base (stocs.cpp:771-786): same pairs in the same order
        pos = orc.scene_centred()
        from oracle import pyoracle
        if nrm is None:
            nrm = pyoracle.normalize_rows(s.nrm)
        k1 = pyoracle.ppf_compute(pos[ids[a][0]], nrm[ids[a][0]], pos[ids[a][1]], nrm[ids[a][1]])
        lo, lg = orc.index_lookup(k1), est.index_lookup(k1)
        assert lo.shape == lg.shape and np.array_equal(lo, lg)
    # congruent sets: pick the two valid bases with the fewest quads to keep the oracle's std::set small
    est.find_congruent_all()
    counts = [(len(est.get_quads(sl)) if False else 0) for sl in range(int(valid.sum()))]
    import ctypes as C
    from model_matching_amd import capi
    sizes = []
    for sl in range(int(valid.sum())):
        n = C.c_int64(0)
        capi.check(est.L.stocs_get_quads(est.h, sl, None, 0, C.byref(n)))
        sizes.append(n.value)
    order = np.argsort(sizes)
    slots_to_attempt = np.nonzero(valid)[0]
    checked = 0
    for sl in order:
        if sizes[sl] == 0 or sizes[sl] > 1500000:
            continue
        a = slots_to_attempt[sl]
        qo = orc.find_congruent(ids[a], float(inv[a][0]), float(inv[a][1]))
        qg = est.get_quads(int(sl))
        assert qo.shape == qg.shape and np.array_equal(qo, qg), (a, qo.shape, qg.shape)
        for q in qo[:: max(1, len(qo) // 50)]:
            oko, To, Po = orc.rigid_transform(ids[a], q)
            okg, Tg, Pg = est.get_rigid_transform_from_congruent_pair(ids[a], q)
            assert oko == okg and (not oko or (np.array_equal(To, Tg) and np.array_equal(Po, Pg)))
        checked += 1
        if checked == 2:
            break
    assert checked == 2


def test_on_demand_quads_at_metric_size():
    """Cm: the count-only congruent pass against its own materialisation (size-independent properties, no oracle):
    per-base counts add up; for a mid-size base the walk order is a permutation of the sorted std::set order;
    for a big base, sampled ranks resolve to distinct members of the materialised set; the candidates of
    make_transforms come from exactly those quads."""
    m, s, est, orc = _pair("Cm")
    seed = 4321
    valid, ids, inv = est.sample_bases(seed, 24)
    nb = int(valid.sum())
    total = est.find_congruent_all()
    sizes = np.array([est.num_quads(b) for b in range(nb)])
    assert sizes.sum() == total and total > 10 ** 6

    def pack(q):
        q = q.astype(np.int64)
        return ((q[:, 0] * 65536 + q[:, 1]) * 65536 + q[:, 2]) * 65536 + q[:, 3]

    mid = [b for b in range(nb) if 1000 <= sizes[b] <= 300000]
    assert mid
    b = mid[0]
    full = est.get_quads(b)
    assert len(full) == sizes[b] and np.all(np.diff(pack(full)) > 0)           # sorted, no duplicates
    emitted = est.get_quads_at(b, np.arange(sizes[b]))
    assert np.array_equal(np.sort(pack(emitted)), pack(full))                   # same set, other order
    big = int(np.argmax(sizes))
    if sizes[big] <= 4 * 10 ** 6:
        fullb = pack(est.get_quads(big))
        rng = np.random.default_rng(1)
        ranks = rng.choice(sizes[big], size=2000, replace=False)
        got = pack(est.get_quads_at(big, ranks))
        assert np.isin(got, fullb).all() and len(np.unique(got)) == len(got)
    # every candidate is the transform of (its base, one of that base's quads): check through the quad counts
    nc = est.make_transforms(200, seed)
    T, P, l, bidx = est.get_pose_candidates()
    assert nc == len(T) and nc <= np.minimum(sizes, 200).sum()
    per_base = np.bincount(bidx, minlength=nb)
    assert np.all(per_base <= np.minimum(sizes, 200))


def test_ppf_float_filter_never_changes_a_key():
    """The pass kernels key point pairs with a float evaluation (own atan2, bin indices without integer division) and fall
    back to the reference's double arithmetic within 1e-3 degree of a bin boundary (sample.hip, ppf_key_fast).  Device
    self-check on seeded pairs of the Cm scene, of a scene with axis-aligned normals (exact 0 / 90 / 180 degree angles sit
    ON integer angles) and of random clouds at several scales and discretisations: a pair the filter calls certain must
    have the key of the double arithmetic -- no exception in 50 million."""
    import ctypes as C
    from model_matching_amd import capi, synth
    from model_matching_amd.estimator import StocsEstimator
    m, s, k = synth.workload("Cm")
    nrm2 = np.zeros_like(s.nrm); nrm2[np.arange(len(s.nrm)), np.arange(len(s.nrm)) % 3] = 1.0
    rng = np.random.default_rng(5)
    cases = [(s.pos, s.nrm, 5, 5, 20_000_000), (s.pos, nrm2, 5, 5, 2_000_000)]
    for scale, tr, rot in ((0.05, 5, 5), (0.3, 1, 1), (1.0, 3, 2), (3.0, 10, 9), (0.1, 7, 10), (0.02, 2, 3), (0.2, 5, 180)):
        n = 20000
        pos = (rng.normal(size=(n, 3)) * scale).astype(np.float32)
        nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        # a share of the normals snapped to exactly representable directions, positions to a lattice: ties and exact angles
        snap = rng.random(n) < 0.2
        nrm[snap] = np.round(nrm[snap]); bad = np.linalg.norm(nrm, axis=1) == 0; nrm[bad] = (0, 0, 1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        pos[snap] = np.round(pos[snap] / (scale / 8)) * np.float32(scale / 8)
        cases.append((pos, nrm.astype(np.float32), tr, rot, 4_000_000))
    total = 0
    for pos, nrm, tr, rot, n_pairs in cases:
        prm = capi.default_params()
        prm.ppf_tr_discretization = tr; prm.ppf_rot_discretization = rot
        est = StocsEstimator(pos, nrm, np.full(len(pos), 0.5, np.float32), None, m.pos[:200], m.nrm[:200], params=prm, build_index=False)
        nt, nu, nm = C.c_int64(), C.c_int64(), C.c_int64()
        assert est.L.stocs_ppf_filter_check(est.h, 99, n_pairs, C.byref(nt), C.byref(nu), C.byref(nm)) == 0
        assert nm.value == 0, (tr, rot, nm.value)
        assert nt.value > 0.99 * n_pairs
        if nrm is s.nrm:
            assert nu.value < 0.01 * nt.value       # the filter decides all but ~0.1 % of generic pairs
        total += nt.value
        est.close()
    assert total > 45_000_000


def test_fixed_point_weight_of_the_draws_all_floats():
    """The seeded draws weigh a point by trunc(w * 2^32); the kernels read it off the float's bits (no double, no branches).
    Device self-check over all 2^32 float patterns against (uint64_t)((double)w * 2^32)."""
    import ctypes as C
    from model_matching_amd import synth
    from model_matching_amd.estimator import StocsEstimator
    m, s, k = synth.workload("tiny")
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    nm = C.c_int64(-1)
    assert est.L.stocs_weight_fix_check(est.h, C.byref(nm)) == 0
    assert nm.value == 0
    est.close()
