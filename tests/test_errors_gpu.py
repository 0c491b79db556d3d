"""Error behaviour of the C ABI on the GPU box: status codes instead of exceptions/crashes, the
call-order contract, capacity reporting, degenerate inputs (the edge cases a caller of the reference
class can hit: no valid base, no congruent set, no candidate, all-zero scores)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(build_index=True, prob=None, n_scene=600, n_model=150):
    from model_matching_amd import synth
    from model_matching_amd.estimator import StocsEstimator
    m = synth.make_model(n_model, seed=21)
    s = synth.make_scene(m, n_scene, seed=22)
    p = s.prob if prob is None else np.full(len(s.pos), prob, np.float32)
    return m, s, StocsEstimator(s.pos, s.nrm, p, s.pixel, m.pos, m.nrm, build_index=build_index)


def test_invalid_arguments_are_status_codes():
    from model_matching_amd import capi, synth
    L = capi.load()
    m = synth.make_model(100, seed=1)
    pos, pp = capi.f32(m.pos); nrm, pn = capi.f32(m.nrm); w, pw = capi.f32(np.ones(100))
    h = C.c_void_p()
    prm = capi.default_params()
    assert L.stocs_ctx_create(C.byref(prm), pp, pn, pw, None, 0, pp, pn, 100, 0, -1, C.byref(h)) == -1      # empty scene
    assert L.stocs_ctx_create(C.byref(prm), pp, pn, pw, None, 100, pp, pn, 0, 0, -1, C.byref(h)) == -1      # empty model
    assert L.stocs_ctx_create(C.byref(prm), None, pn, pw, None, 100, pp, pn, 100, 0, -1, C.byref(h)) == -1   # NULL cloud
    assert L.stocs_ctx_create(C.byref(prm), pp, pn, pw, None, 100, pp, pn, 70000, 0, -1, C.byref(h)) == -1   # > 65535 model points
    bad = capi.default_params(ppf_rot_discretization=7)
    assert L.stocs_ctx_create(C.byref(bad), pp, pn, pw, None, 100, pp, pn, 100, 0, -1, C.byref(h)) == -1
    assert L.stocs_ctx_create(C.byref(prm), pp, pn, pw, None, 100, pp, pn, 100, 0, 99, C.byref(h)) == -2     # no such device
    assert not h.value and len(L.stocs_last_error()) > 0
    assert L.stocs_ctx_destroy(None) == 0 and L.stocs_sync(None) == -1


def test_call_order_contract():
    from model_matching_amd import capi
    m, s, est = _mk(build_index=False)
    with pytest.raises(capi.StocsError) as e:
        est.sample_bases(1, 4)
    assert e.value.code == -5                                    # STATE: no index
    with pytest.raises(capi.StocsError):
        est.index_lookup((50, 90, 90, 0))
    with pytest.raises(capi.StocsError):
        est.find_congruent_all()
    # scoring works without an index
    assert est.score_transforms(np.eye(4, dtype=np.float32).reshape(1, 16)).shape == (1,)
    m, s, est = _mk(build_index=True)
    with pytest.raises(capi.StocsError) as e:
        est.make_transforms(200, 1)                              # before find_congruent_all
    assert e.value.code == -5
    with pytest.raises(capi.StocsError):
        est.get_quads(0)
    assert est.find_congruent_all() == 0                         # no bases: nothing to do, not an error
    assert est.make_transforms(200, 1) == 0
    assert est.compute_best_transform()[:2] == (0.0, -1)         # no candidates: best_index -1 (NULL best pose)
    with pytest.raises(capi.StocsError):
        est.set_bases(np.array([[0, 1, 2, 10 ** 6]]), np.zeros((1, 2)))   # scene index out of range
    with pytest.raises(capi.StocsError):
        est.set_option("no_such_option", 1)


def test_capacity_is_reported_not_overrun():
    from model_matching_amd import capi
    m, s, est = _mk()
    valid, ids, inv = est.sample_bases(5, 30)
    assert est.find_congruent_all() > 0
    L = capi.load()
    n = C.c_int64(0)
    slot = next(i for i in range(int(valid.sum())) if len(est.get_quads(i)) > 3)
    buf = np.full((2, 4), -7, np.int32)
    rc = L.stocs_get_quads(est.h, slot, buf.ctypes.data_as(capi._ip), 2, C.byref(n))
    assert rc == -4 and n.value > 2 and (buf >= 0).all()         # CAPACITY: first 2 written, needed count returned
    full = est.get_quads(slot)
    assert np.array_equal(full[:2], buf)
    est.make_transforms(200, 5)
    nc = C.c_int(0)
    T = np.zeros((1, 16), np.float32)
    rc = L.stocs_get_candidates(est.h, T.ctypes.data_as(capi._fp), None, None, None, 1, C.byref(nc))
    assert nc.value > 1 and rc == -4


def test_degenerate_scenes():
    # class probability zero everywhere: every attempt fails on its first draw ("Zero probability returned")
    m, s, est = _mk(prob=0.0)
    valid, ids, inv = est.sample_bases(3, 10)
    assert not valid.any() and est.L.stocs_num_bases(est.h) == 0
    # a model far bigger than anything in the scene: bases exist, no congruent sets, no candidates, no pose
    from model_matching_amd import synth
    from model_matching_amd.estimator import StocsEstimator
    big = synth.make_model(150, seed=5, scale=8.0)
    sc = synth.make_scene(synth.make_model(150, seed=21), 600, seed=22)
    est = StocsEstimator(sc.pos, sc.nrm, sc.prob, sc.pixel, big.pos, big.nrm, build_index=True)
    valid, ids, inv = est.sample_bases(3, 20)
    nq = est.find_congruent_all()
    nc = est.make_transforms(200, 3)
    lcp, idx, pose = est.compute_best_transform()
    assert nc <= nq and (idx == -1) == (lcp == 0.0)
    if idx == -1:
        assert not pose.any()


def test_native_rccl_allreduce_single_rank():
    """stocs_allreduce_best over RCCL with a one-rank communicator (the only size a one-GPU box allows):
    the key and pose come back unchanged; the all-zero key means "no pose" (Q18)."""
    from model_matching_amd import capi
    L = capi.load()
    uid = (C.c_char * 128)()
    assert L.stocs_comm_unique_id(uid) == 0, L.stocs_last_error()
    comm = C.c_void_p()
    assert L.stocs_comm_create(uid, 1, 0, 0, C.byref(comm)) == 0, L.stocs_last_error()
    key = C.c_uint64(L.stocs_pack_best(C.c_float(0.375), 1234))
    pose = np.arange(16, dtype=np.float32)
    p2 = pose.copy()
    assert L.stocs_allreduce_best(comm, None, C.byref(key), p2.ctypes.data_as(capi._fp), 65536) == 0, L.stocs_last_error()
    s, i = C.c_float(), C.c_uint32()
    L.stocs_unpack_best(key.value, C.byref(s), C.byref(i))
    assert (s.value, i.value) == (0.375, 1234) and np.array_equal(p2, pose)
    key = C.c_uint64(0)
    assert L.stocs_allreduce_best(comm, None, C.byref(key), p2.ctypes.data_as(capi._fp), 65536) == 0
    assert key.value == 0 and not p2.any()
    # a rank whose candidates all scored 0 contributes "none", not a pose with lcp 0 (stocs.cpp:987-998)
    assert L.stocs_pack_best(C.c_float(0.0), 17) == 0 and L.stocs_pack_best(C.c_float(-1.0), 17) == 0
    assert L.stocs_pack_best(C.c_float(float("nan")), 17) == 0
    key = C.c_uint64(L.stocs_pack_best(C.c_float(0.5), 70000))      # id maps to rank 1 of a 1-rank job
    assert L.stocs_allreduce_best(comm, None, C.byref(key), p2.ctypes.data_as(capi._fp), 65536) == -1
    assert L.stocs_comm_destroy(comm) == 0
    assert L.stocs_comm_create(uid, 2, 5, 0, C.byref(comm)) == -1   # rank out of range


def test_on_demand_quads_edge_cases(oracle_lib):
    """stocs_get_quads_at range checks; per-base maximum of 1 (every non-empty base is sampled, none materialised)
    and a huge maximum (every base is materialised and used whole) against the oracle; stream switch keeps state."""
    from model_matching_amd import capi
    m, s, est = _mk()
    orc = oracle_lib.Oracle(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=True)
    seed = 5
    valid, ids, inv = est.sample_bases(seed, 30)
    assert est.find_congruent_all() > 0
    slot = next(i for i in range(int(valid.sum())) if est.num_quads(i) > 3)
    nq = est.num_quads(slot)
    for bad in ([-1], [nq], [0, nq + 5]):
        with pytest.raises(capi.StocsError) as e:
            est.get_quads_at(slot, bad)
        assert e.value.code == -1
    with pytest.raises(capi.StocsError):
        est.get_quads_at(10 ** 6, [0])
    assert est.get_quads_at(slot, []).shape == (0, 4)
    # same rank twice is allowed, and equals the oracle's walk-order sequence
    a = int(np.nonzero(valid)[0][slot])
    so = orc.find_congruent_seq(ids[a], float(inv[a][0]), float(inv[a][1]))
    assert np.array_equal(est.get_quads_at(slot, [2, 2, 0]), so[[2, 2, 0]])
    for max_sets in (1, 10 ** 6):
        r = orc.run(seed, 30, max_sets)
        assert est.make_transforms(max_sets, seed) == r.n_candidates
        To, Po, bo = orc.candidates()
        Tg, Pg, lg, bg = est.get_pose_candidates()
        assert np.array_equal(To, Tg) and np.array_equal(bo, bg)
        lcp, idx, pose = est.compute_best_transform()
        assert abs(lcp - r.best_lcp) <= 1e-5
    with pytest.raises(capi.StocsError):
        est.make_transforms((1 << 24) + 1, seed)
    # switching the stream keeps the counted state usable
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    est.set_stream(st.value)
    assert np.array_equal(est.get_quads_at(slot, [1]), so[[1]])
    est.set_stream(None)
    assert hip.hipStreamDestroy(st) == 0
    assert np.array_equal(est.get_quads(slot), orc.find_congruent(ids[a], float(inv[a][0]), float(inv[a][1])))
    est.close()


def test_set_scene_and_workspace_calls_reject_bad_arguments():
    from model_matching_amd import capi
    from model_matching_amd.estimator import ingest_scene
    L = capi.load()
    m, s, est = _mk()
    pos, pp = capi.f32(s.pos); nrm, pn = capi.f32(s.nrm); pr, ppr = capi.f32(s.prob)
    assert L.stocs_ctx_set_scene(None, pp, pn, ppr, None, len(pos)) == -1
    assert L.stocs_ctx_set_scene(est.h, pp, pn, ppr, None, 0) == -1
    assert L.stocs_ctx_set_scene(est.h, None, pn, ppr, None, len(pos)) == -1
    assert b"set_scene" in L.stocs_last_error()
    # a failed call leaves the context usable with its old scene
    v, ids, inv = est.sample_bases(3, 10)
    assert len(v) == 10
    # pixels outside the image of stocs_params are refused (instance mode indexes its 2-D maps with them), at creation and
    # per frame; the refused call changes nothing
    h = C.c_void_p()
    prm = capi.default_params()
    mp, pmp = capi.f32(m.pos); mn, pmn = capi.f32(m.nrm)
    for bad_px in ((480, 0), (0, 640), (-1, 5), (5, -1)):
        px = np.ascontiguousarray(s.pixel, np.int32).copy()
        px[7] = bad_px
        ppx = px.ctypes.data_as(capi._ip)
        assert L.stocs_ctx_set_scene(est.h, pp, pn, ppr, ppx, len(pos)) == -1 and b"pixel" in L.stocs_last_error()
        assert L.stocs_ctx_create(C.byref(prm), pp, pn, ppr, ppx, len(pos), pmp, pmn, len(mp), 0, -1, C.byref(h)) == -1 and not h.value
    v1, ids1, inv1 = est.sample_bases(3, 10)
    assert np.array_equal(v, v1) and np.array_equal(ids[v], ids1[v1])
    # pixels are optional (class mode never reads them)
    assert L.stocs_ctx_set_scene(est.h, pp, pn, ppr, None, len(pos)) == 0
    v2, ids2, inv2 = est.sample_bases(3, 10)
    assert np.array_equal(v, v2) and np.array_equal(ids[v], ids2[v2])
    # the ingest workspace of this thread can be given back at any time and is rebuilt on demand
    depth = np.full((48, 64), 8000, np.uint16); prob = np.full((48, 64), 9000, np.uint16)
    a = ingest_scene(depth, prob, (60.0, 32.0, 60.0, 24.0), 1e-4)
    assert L.stocs_trim() == 0 and L.stocs_trim() == 0
    b = ingest_scene(depth, prob, (60.0, 32.0, 60.0, 24.0), 1e-4)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    est.close()


def test_patch_test_entry_points_reject_bad_arguments():
    """stocs_get_cull_state / stocs_model_patch_order / the options behind the patch test and the verify trips: status codes, and a
    model too small for the test (fewer than 64 points) simply has no distance field."""
    from model_matching_amd import capi
    L = capi.load()
    m, s, est = _mk(build_index=False, n_scene=600, n_model=150)
    npat, nd = C.c_int(-1), C.c_int64(-1)
    assert L.stocs_get_cull_state(None, None, None, C.byref(npat), None, None, 0, C.byref(nd)) == -1
    assert L.stocs_get_cull_state(est.h, None, None, None, None, None, 0, C.byref(nd)) == -1
    assert L.stocs_get_cull_state(est.h, None, None, C.byref(npat), None, None, 0, C.byref(nd)) == 0
    assert npat.value == 3 and nd.value > 0
    small = np.zeros(8, np.float32)
    assert L.stocs_get_cull_state(est.h, None, None, C.byref(npat), None, small.ctypes.data_as(capi._fp), 8, C.byref(nd)) == -4   # capacity
    for key, bad in (("lcp_cull", 3), ("lcp_cull", -1), ("lcp_group", 3), ("lcp_group", 8), ("lcp_group", 16), ("lcp_order", -1)):   # (8 lanes per query: measurement build only)
        assert L.stocs_set_option(est.h, key.encode(), bad) == -1, (key, bad)
    for key, ok in (("lcp_cull", 0), ("lcp_cull", 2), ("lcp_group", 4)):
        assert L.stocs_set_option(est.h, key.encode(), ok) == 0, (key, ok)
    est.close()
    # fewer than 64 model points: no field, scoring works as ever
    m2, s2, est2 = _mk(build_index=False, n_scene=600, n_model=40)
    assert L.stocs_get_cull_state(est2.h, None, None, C.byref(npat), None, None, 0, C.byref(nd)) == 0 and nd.value == 0
    est2.set_option("lcp_cull", 2)
    T = np.eye(4, dtype=np.float32).reshape(1, 16)
    assert est2.score_transforms(T).shape == (1,)
    est2.close()
    perm = np.zeros(10, np.int32); pat = np.zeros(4, np.float32); pos = np.zeros(30, np.float32)
    assert L.stocs_model_patch_order(None, 10, perm.ctypes.data_as(capi._ip), pat.ctypes.data_as(capi._fp)) == -1
    assert L.stocs_model_patch_order(pos.ctypes.data_as(capi._fp), 0, perm.ctypes.data_as(capi._ip), pat.ctypes.data_as(capi._fp)) == -1
    assert L.stocs_model_patch_order(pos.ctypes.data_as(capi._fp), 10, perm.ctypes.data_as(capi._ip), pat.ctypes.data_as(capi._fp)) == 0
    assert sorted(perm.tolist()) == list(range(10)) and pat[3] >= 0.0    # ten coincident points: a sphere of radius ~0
