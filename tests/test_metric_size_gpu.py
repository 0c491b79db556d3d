"""The whole path at the METRIC size (SURVEY 8d "Cm": 20 000-point scene, 5 000-point model, 100 base attempts, <= 200 congruent sets
per base) -- the sizes at which the congruent phase's capacities, the trial batch's natural cut into pieces and the arena ceiling
actually bind (reference: run_stocs_estimation, src/stocs_match_one_object.cpp:81-165; find_congruent_sets_on_model, src/stocs.cpp:753-869).

tests/test_trials_gpu.py and tests/test_pipeline_gpu.py pin the same mechanisms on the small workloads, where every piece is forced
by an environment knob; here nothing is forced except where a test says so."""
import numpy as np
import pytest

from test_trials_gpu import _single

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cm_est():
    from model_matching_amd import synth
    from model_matching_amd.estimator import StocsEstimator
    m, s, k = synth.workload("Cm")
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=True)
    yield m, s, est
    est.close()


def _trial_equals_single(est, res, t, seed, n_attempts=100, max_per_base=200):
    ref = _single(est, int(seed), n_attempts, 0, max_per_base)
    g = res[t]
    assert (g["n_bases"], g["n_quads"], g["n_candidates"]) == (int(ref["valid"].sum()), ref["n_quads"], ref["n_candidates"]), t
    assert g["best_index"] == ref["best_index"] and g["best_lcp"] == ref["best_lcp"], t
    assert np.array_equal(g["best_pose"].view(np.uint32), ref["best_pose"].view(np.uint32)), t
    return ref


def test_a_plan_beyond_its_capacity_at_the_metric_size(cm_est, monkeypatch):
    """stocs_find_congruent_all sizes buffers and launches of the second and later trials of a scene by a capacity (1.6 x the trial
    before); every kernel launched under it bounds its indices by the CAPACITY, and a plan that turns out larger is redone with exact
    sizes.  Forced here at full size: a capacity of HALF the previous trial's lists (~4-5 M entries short per list) -- the case in
    which round 4's survivors_base_offsets_kernel followed the planned offsets past the gathered keys."""
    m, s, est = cm_est
    est.reset_trial()
    valid, ids, inv = est.sample_bases(4243, 100)
    nv = int(valid.sum())
    monkeypatch.setenv("STOCS_CONGRUENT_EXACT_SIZES", "1")
    n_exact = est.find_congruent_all()
    counts = np.array([est.num_quads(k) for k in range(nv)], np.int64)
    nc = est.make_transforms(200, 4243)
    T0 = est.get_pose_candidates()[0].copy()
    monkeypatch.delenv("STOCS_CONGRUENT_EXACT_SIZES")
    assert n_exact > 10**6 and nc > 1000
    for capacity, redone in (("0.5", True), ("0.98", None), (None, False)):
        if capacity:
            monkeypatch.setenv("STOCS_CONGRUENT_CAPACITY", capacity)
        assert est.find_congruent_all() == n_exact
        steps = [lab for lab, _ in est.last_call_timing(0)]
        if redone is not None:
            assert ("plan beyond the capacities: redone with exact sizes" in steps) == redone, (capacity, steps)
        assert np.array_equal(np.array([est.num_quads(k) for k in range(nv)], np.int64), counts)
        assert est.make_transforms(200, 4243) == nc
        assert np.array_equal(est.get_pose_candidates()[0].view(np.uint32), T0.view(np.uint32))
        if capacity:
            monkeypatch.delenv("STOCS_CONGRUENT_CAPACITY")
    # a DIFFERENT base set under a capacity learnt from this one, short by a fifth: whichever way it falls, the sets are the exact ones
    est.reset_trial()
    v2, _, _ = est.sample_bases(77, 100)
    monkeypatch.setenv("STOCS_CONGRUENT_EXACT_SIZES", "1")
    n2 = est.find_congruent_all()
    c2 = np.array([est.num_quads(k) for k in range(int(v2.sum()))], np.int64)
    monkeypatch.delenv("STOCS_CONGRUENT_EXACT_SIZES")
    est.reset_trial(); est.sample_bases(4243, 100); est.find_congruent_all()        # history: the first base set
    est.reset_trial(); est.sample_bases(77, 100)
    monkeypatch.setenv("STOCS_CONGRUENT_CAPACITY", "0.8")
    assert est.find_congruent_all() == n2
    assert np.array_equal(np.array([est.num_quads(k) for k in range(int(v2.sum()))], np.int64), c2)


def test_a_batch_at_the_metric_size_is_cut_where_the_quad_keys_end(cm_est, monkeypatch):
    """44 trials of 100 attempts at Cm: the packed 64-bit quads hold 4 x 13 id bits + 12 base bits = 40 trials' bases, so the batch runs
    as two natural pieces (40 + 4) -- nothing forced.  Trials on both sides of the cut are bit for bit the same seeds run alone, a second
    identical call allocates nothing, and a memory ceiling that forces the `too_big` halving gives the same trials again."""
    m, s, est = cm_est
    seeds = [9000 + 7 * i for i in range(44)]
    res = est.run_trials(seeds, 100, max_per_base=200)
    tm = dict(est.last_call_timing(3))
    assert tm["pieces (sets of launches) the batch was cut into"] == 2.0, tm
    n_alloc = int(est.L.stocs_device_alloc_count())
    again = est.run_trials(seeds, 100, max_per_base=200)
    assert int(est.L.stocs_device_alloc_count()) == n_alloc                # a warm context runs the batch again without a device allocation
    for a, b in zip(res, again):
        assert (a["n_bases"], a["n_quads"], a["n_candidates"], a["best_index"], a["best_lcp"]) == (b["n_bases"], b["n_quads"], b["n_candidates"], b["best_index"], b["best_lcp"])
    # a ceiling of 6 GB of arena: the 40-trial piece (~36 GB of pair lists) is planned, found too big and halved until it fits
    monkeypatch.setenv("STOCS_TRIALS_MAX_MB", "6144")
    cut = est.run_trials(seeds, 100, max_per_base=200)
    monkeypatch.delenv("STOCS_TRIALS_MAX_MB")
    assert dict(est.last_call_timing(3))["pieces (sets of launches) the batch was cut into"] >= 5.0
    for a, b in zip(res, cut):
        assert (a["n_bases"], a["n_quads"], a["n_candidates"], a["best_index"], a["best_lcp"]) == (b["n_bases"], b["n_quads"], b["n_candidates"], b["best_index"], b["best_lcp"])
        assert np.array_equal(a["best_pose"].view(np.uint32), b["best_pose"].view(np.uint32))
    assert sum(r["n_candidates"] for r in res) > 44 * 3000
    for t in (0, 39, 40, 43):                                              # first and last trial of either piece
        _trial_equals_single(est, res, t, seeds[t])
