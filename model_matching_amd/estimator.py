"""Python mirror of the reference's ``stocs::stocs_estimator`` (reference include/stocs.hpp:16-180)
over the C ABI of libstocs_hip.so.  Same method names and argument meaning as the reference class
(flat clouds instead of PLY/PNG paths); every method forwards to the HIP library -- nothing is
computed in Python."""
from __future__ import annotations

import atexit
import ctypes as C
import weakref

import numpy as np

from . import capi

# contexts still open when the interpreter exits are destroyed before module teardown, i.e. while the HIP runtime
# (possibly shared with PyTorch) is still up; a __del__ that runs after the runtime's own shutdown would free twice
_LIVE = weakref.WeakSet()


@atexit.register
def _close_all():
    for est in list(_LIVE):
        try:
            est.close()
        except Exception:
            pass


# stocs_trial_result as a numpy record (same layout as capi.TrialResult: 4 + 4 + 8 + 4 + 4 + 64 bytes)
_TRIAL_DTYPE = np.dtype([("n_bases", np.int32), ("n_candidates", np.int32), ("n_quads", np.int64), ("best_lcp", np.float32), ("best_index", np.int32),
                         ("best_pose16", np.float32, (16,))])
assert _TRIAL_DTYPE.itemsize == C.sizeof(capi.TrialResult)


class StocsEstimator:
    def __init__(self, scene_pos, scene_nrm, scene_prob, scene_pixel, model_pos, model_nrm,
                 params: capi.Params | None = None, build_index: bool = True, device: int = -1):
        self.L = capi.load()
        self.prm = params or capi.default_params()
        self._sp, psp = capi.f32(scene_pos)
        self._sn, psn = capi.f32(scene_nrm)
        self._spr, pspr = capi.f32(scene_prob)
        ppx = None
        if scene_pixel is not None:
            self._spx, ppx = capi.i32(scene_pixel)
        self._mp, pmp = capi.f32(model_pos)
        self._mn, pmn = capi.f32(model_nrm)
        self.nS, self.nM = len(self._sp), len(self._mp)
        self.h = C.c_void_p()
        capi.check(self.L.stocs_ctx_create(C.byref(self.prm), psp, psn, pspr, ppx, self.nS, pmp, pmn, self.nM,
                                           1 if build_index else 0, device, C.byref(self.h)))
        _LIVE.add(self)

    def set_scene(self, scene_pos, scene_nrm, scene_prob, scene_pixel=None):
        """The next camera frame against the same model (stocs_ctx_set_scene): keeps the model and its PPF index."""
        self._sp, psp = capi.f32(scene_pos)
        self._sn, psn = capi.f32(scene_nrm)
        self._spr, pspr = capi.f32(scene_prob)
        ppx = None
        if scene_pixel is not None:
            self._spx, ppx = capi.i32(scene_pixel)
        capi.check(self.L.stocs_ctx_set_scene(self.h, psp, psn, pspr, ppx, len(self._sp)))
        self.nS = len(self._sp)

    def close(self):
        _LIVE.discard(self)
        if getattr(self, "h", None) and self.h.value:
            self.L.stocs_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (stocs.hpp:115-134) ----
    def get_scene_centroid(self):
        s = np.zeros(3, np.float32)
        capi.check(self.L.stocs_get_centroids(self.h, s.ctypes.data_as(capi._fp), None))
        return s

    def get_model_centroid(self):
        m = np.zeros(3, np.float32)
        capi.check(self.L.stocs_get_centroids(self.h, None, m.ctypes.data_as(capi._fp)))
        return m

    def set_edge_map(self, edge):
        e = np.ascontiguousarray(edge, np.uint8)
        capi.check(self.L.stocs_set_edge_map(self.h, e.ctypes.data_as(capi._u8p)))

    def reset_trial(self):
        capi.check(self.L.stocs_reset_trial(self.h))

    # ---- PPF index ----
    def index_exists(self, key):
        k, pk = capi.i32(key)
        r = C.c_int(0)
        capi.check(self.L.stocs_index_exists(self.h, pk, C.byref(r)))
        return bool(r.value)

    def index_lookup(self, key):
        k, pk = capi.i32(key)
        n = C.c_int64(0)
        capi.check(self.L.stocs_index_lookup(self.h, pk, None, 0, C.byref(n)))
        out = np.zeros((n.value, 2), np.int32)
        if n.value:
            capi.check(self.L.stocs_index_lookup(self.h, pk, out.ctypes.data_as(capi._ip), n.value, C.byref(n)))
        return out

    def index_save(self, path):
        capi.check(self.L.stocs_index_save(self.h, str(path).encode()))

    def index_load(self, path):
        capi.check(self.L.stocs_index_load(self.h, str(path).encode()))

    def index_stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        capi.check(self.L.stocs_index_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- base sampling (stocs.hpp:80-91) ----
    def sample_bases(self, seed, n_attempts, first_attempt=0, mode=0, dispersion=0.9):
        ids = np.zeros((n_attempts, 4), np.int32)
        inv = np.zeros((n_attempts, 2), np.float32)
        valid = np.zeros(n_attempts, np.int32)
        capi.check(self.L.stocs_sample_bases(self.h, mode, seed, first_attempt, n_attempts, dispersion,
                                             ids.ctypes.data_as(capi._ip), inv.ctypes.data_as(capi._fp),
                                             valid.ctypes.data_as(capi._ip)))
        return valid.astype(bool), ids, inv

    def sample_class_base(self, seed, attempt):
        v, ids, inv = self.sample_bases(seed, 1, attempt, 0)
        return bool(v[0]), ids[0], inv[0]

    def set_bases(self, ids, inv):
        ids, pi = capi.i32(np.asarray(ids).reshape(-1, 4))
        inv, pv = capi.f32(np.asarray(inv).reshape(-1, 2))
        capi.check(self.L.stocs_set_bases(self.h, len(ids), pi, pv))

    def class_pass(self, k, b3, w_in):
        b, pb = capi.i32(b3)
        w, pw = capi.f32(w_in)
        out = np.zeros_like(w)
        capi.check(self.L.stocs_class_pass(self.h, k, pb, pw, out.ctypes.data_as(capi._fp)))
        return out

    def try_sampled_base(self, ids):
        ids, pi = capi.i32(np.array(ids).copy())
        inv = np.zeros(2, np.float32)
        ok = C.c_int(0)
        capi.check(self.L.stocs_try_sampled_base(self.h, pi, inv.ctypes.data_as(capi._fp), C.byref(ok)))
        return bool(ok.value), ids, inv

    def draw(self, w, r64):
        w, pw = capi.f32(w)
        idx = C.c_int(0)
        capi.check(self.L.stocs_draw(self.h, pw, len(w), r64, C.byref(idx)))
        return idx.value

    # ---- congruent sets (stocs.hpp:93-96) ----
    def find_congruent_all(self):
        n = C.c_int64(0)
        capi.check(self.L.stocs_find_congruent_all(self.h, C.byref(n)))
        return n.value

    def get_quads(self, slot):
        n = C.c_int64(0)
        capi.check(self.L.stocs_get_quads(self.h, slot, None, 0, C.byref(n)))
        out = np.zeros((n.value, 4), np.int32)
        if n.value:
            capi.check(self.L.stocs_get_quads(self.h, slot, out.ctypes.data_as(capi._ip), n.value, C.byref(n)))
        return out

    def num_quads(self, slot):
        n = C.c_int64(0)
        capi.check(self.L.stocs_get_quads(self.h, slot, None, 0, C.byref(n)))
        return n.value

    def get_quads_at(self, slot, ranks):
        """Quads of base `slot` at ranks of its walk order (stocs_get_quads_at)."""
        r = np.ascontiguousarray(ranks, np.int64)
        out = np.zeros((len(r), 4), np.int32)
        capi.check(self.L.stocs_get_quads_at(self.h, slot, r.ctypes.data_as(capi._i64p), len(r), out.ctypes.data_as(capi._ip)))
        return out

    def find_congruent_sets_on_model(self, base_indices, invariant1, invariant2):
        self.set_bases(np.asarray(base_indices).reshape(1, 4), np.array([[invariant1, invariant2]], np.float32))
        self.find_congruent_all()
        return self.get_quads(0)

    # ---- transforms (stocs.hpp:98-101) ----
    def get_rigid_transform_from_congruent_pair(self, base_indices, quad):
        ids, pi = capi.i32(base_indices)
        q, pq = capi.i32(quad)
        T = np.zeros(16, np.float32); P = np.zeros(16, np.float32)
        ok = C.c_int(0)
        capi.check(self.L.stocs_rigid_transform(self.h, pi, pq, T.ctypes.data_as(capi._fp), P.ctypes.data_as(capi._fp), C.byref(ok)))
        return bool(ok.value), T, P

    def make_transforms(self, max_per_base=200, seed=0):
        n = C.c_int(0)
        capi.check(self.L.stocs_make_transforms(self.h, max_per_base, seed, C.byref(n)))
        return n.value

    def get_pose_candidates(self):
        n = C.c_int(0)
        capi.check(self.L.stocs_get_candidates(self.h, None, None, None, None, 0, C.byref(n)))
        T = np.zeros((n.value, 16), np.float32); P = np.zeros((n.value, 16), np.float32)
        l = np.zeros(n.value, np.float32); b = np.zeros(n.value, np.int32)
        if n.value:
            capi.check(self.L.stocs_get_candidates(self.h, T.ctypes.data_as(capi._fp), P.ctypes.data_as(capi._fp),
                                                   l.ctypes.data_as(capi._fp), b.ctypes.data_as(capi._ip), n.value, C.byref(n)))
        return T, P, l, b

    # ---- verification (stocs.hpp:103-107) ----
    def compute_alignment_score_for_rigid_transform(self, T16):
        return float(self.score_transforms(np.asarray(T16, np.float32).reshape(1, 16))[0])

    def score_transforms(self, T16):
        T, pT = capi.f32(T16)
        n = T.size // 16
        out = np.zeros(n, np.float32)
        capi.check(self.L.stocs_score_transforms(self.h, pT, n, out.ctypes.data_as(capi._fp)))
        return out

    def lcp_detail(self, T16):
        T, pT = capi.f32(T16)
        hit = np.zeros(self.nM, np.int32); counted = np.zeros(self.nM, np.uint8)
        capi.check(self.L.stocs_lcp_detail(self.h, pT, hit.ctypes.data_as(capi._ip), counted.ctypes.data_as(capi._u8p)))
        return hit, counted

    def lcp_hit_count(self, dT, n):
        """(hits, counted) over n device-resident transforms (stocs_lcp_hit_count): point queries that found a scene point within
        epsilon, and those of them that passed the normal test."""
        h = C.c_int64(0); k = C.c_int64(0)
        capi.check(self.L.stocs_lcp_hit_count(self.h, dT, n, C.byref(h), C.byref(k)))
        return h.value, k.value

    def compute_best_transform(self):
        s = C.c_float(0); i = C.c_int(-1)
        P = np.zeros(16, np.float32)
        capi.check(self.L.stocs_verify_all(self.h, C.byref(s), C.byref(i), P.ctypes.data_as(capi._fp)))
        self.best_lcp, self.best_index, self.best_pose = s.value, i.value, P
        return s.value, i.value, P

    # ---- trial batches: N independent trials in one set of launches (stocs_run_trials) ----
    def run_trials(self, seeds, n_attempts=100, mode=0, dispersion=0.9, max_per_base=200, keep_details=False):
        """-> list of dicts (n_bases, n_candidates, n_quads, best_lcp, best_index, best_pose 16 floats), one per seed."""
        sd = np.ascontiguousarray(seeds, np.uint64)
        res = (capi.TrialResult * max(len(sd), 1))()
        capi.check(self.L.stocs_run_trials(self.h, mode, len(sd), sd.ctypes.data_as(C.POINTER(C.c_uint64)), n_attempts, dispersion, max_per_base,
                                           1 if keep_details else 0, res))
        # (one view of the whole result array: a ctypes field access per trial costs ~20 us, and a thousand trials take 60 ms of device time)
        a = np.frombuffer(res, dtype=_TRIAL_DTYPE, count=len(sd))
        nb, nc, nq, bl, bi, bp = (a["n_bases"].tolist(), a["n_candidates"].tolist(), a["n_quads"].tolist(), a["best_lcp"].tolist(), a["best_index"].tolist(),
                                  a["best_pose16"].copy())
        return [dict(n_bases=nb[t], n_candidates=nc[t], n_quads=nq[t], best_lcp=bl[t], best_index=bi[t], best_pose=bp[t]) for t in range(len(sd))]

    def trial_bases(self, trial):
        n = C.c_int(0)
        capi.check(self.L.stocs_trials_get_bases(self.h, trial, None, None, None, 0, C.byref(n)))
        ids = np.zeros((n.value, 4), np.int32); inv = np.zeros((n.value, 2), np.float32); valid = np.zeros(n.value, np.int32)
        capi.check(self.L.stocs_trials_get_bases(self.h, trial, ids.ctypes.data_as(capi._ip), inv.ctypes.data_as(capi._fp), valid.ctypes.data_as(capi._ip),
                                                 n.value, C.byref(n)))
        return valid.astype(bool), ids, inv

    def trial_quad_counts(self, trial):
        n = C.c_int(0)
        capi.check(self.L.stocs_trials_get_quad_counts(self.h, trial, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int64)
        if n.value:
            capi.check(self.L.stocs_trials_get_quad_counts(self.h, trial, out.ctypes.data_as(capi._i64p), n.value, C.byref(n)))
        return out

    def trial_candidates(self, trial):
        n = C.c_int(0)
        capi.check(self.L.stocs_trials_get_candidates(self.h, trial, None, None, None, None, 0, C.byref(n)))
        T = np.zeros((n.value, 16), np.float32); P = np.zeros((n.value, 16), np.float32)
        l = np.zeros(n.value, np.float32); b = np.zeros(n.value, np.int32)
        if n.value:
            capi.check(self.L.stocs_trials_get_candidates(self.h, trial, T.ctypes.data_as(capi._fp), P.ctypes.data_as(capi._fp), l.ctypes.data_as(capi._fp),
                                                          b.ctypes.data_as(capi._ip), n.value, C.byref(n)))
        return T, P, l, b

    # ---- device-resident scoring for the benchmark ----
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        capi.check(self.L.stocs_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        capi.check(self.L.stocs_dev_free(self.h, p))

    def dev_upload(self, p, arr):
        arr = np.ascontiguousarray(arr)
        capi.check(self.L.stocs_dev_upload(self.h, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def dev_download(self, p, arr):
        capi.check(self.L.stocs_dev_download(self.h, arr.ctypes.data_as(C.c_void_p), p, arr.nbytes))

    def score_device(self, dT, n, dL):
        capi.check(self.L.stocs_score_transforms_device(self.h, dT, n, dL))

    def get_scene(self):
        """The scene as the context holds it (stocs_get_scene): centred positions, unit normals, CURRENT class
        probabilities (instance-mode sampling decays them, Q8), pixels."""
        pos = np.zeros((self.nS, 3), np.float32); nrm = np.zeros((self.nS, 3), np.float32)
        prob = np.zeros(self.nS, np.float32); pix = np.zeros((self.nS, 2), np.int32)
        capi.check(self.L.stocs_get_scene(self.h, pos.ctypes.data_as(capi._fp), nrm.ctypes.data_as(capi._fp), prob.ctypes.data_as(capi._fp),
                                          pix.ctypes.data_as(capi._ip)))
        return pos, nrm, prob, pix

    def get_segment(self):
        """`segment` of the last instance-mode attempt (stocs_get_segment): scene indices."""
        n = C.c_int(0)
        capi.check(self.L.stocs_get_segment(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        if n.value:
            capi.check(self.L.stocs_get_segment(self.h, out.ctypes.data_as(capi._ip), n.value, C.byref(n)))
        return out

    def cull_state(self, with_field=True):
        """(patches (n,4), perm (|M|,), geom dict, dist (nz,ny,nx) or None) of the scoring kernels' patch test (diagnostics)."""
        npat = C.c_int(0); nd = C.c_int64(0)
        capi.check(self.L.stocs_get_cull_state(self.h, None, None, C.byref(npat), None, None, 0, C.byref(nd)))
        patches = np.zeros((max(npat.value, 1), 4), np.float32); perm = np.zeros(self.nM, np.int32); geom = np.zeros(8, np.float32)
        dist = np.zeros(max(nd.value, 1), np.float32) if with_field else None
        capi.check(self.L.stocs_get_cull_state(self.h, patches.ctypes.data_as(capi._fp), perm.ctypes.data_as(capi._ip), C.byref(npat), geom.ctypes.data_as(capi._fp),
                                               dist.ctypes.data_as(capi._fp) if with_field else None, nd.value, C.byref(nd)))
        g = dict(origin=geom[:3].copy(), g=float(geom[3]), cap=float(geom[4]), dims=(int(geom[5]), int(geom[6]), int(geom[7])))
        if with_field and nd.value:
            dist = dist[:nd.value].reshape(g["dims"][2], g["dims"][1], g["dims"][0])
        else:
            dist = None
        return patches[:npat.value], perm, g, dist

    def set_option(self, key, value):
        capi.check(self.L.stocs_set_option(self.h, key.encode(), int(value)))

    def best_device(self, dL, n, id_offset=0):
        """(best_lcp, global id) of n device-resident scores; (0.0, -1) when none is positive."""
        key = C.c_uint64(0)
        capi.check(self.L.stocs_best_device(self.h, dL, n, id_offset, C.byref(key)))
        if key.value == 0:
            return 0.0, -1, 0
        s = C.c_float(0); i = C.c_uint32(0)
        self.L.stocs_unpack_best(key.value, C.byref(s), C.byref(i))
        return s.value, int(i.value), int(key.value)

    def score_best_device(self, dT, n, dL, id_offset=0):
        key = C.c_uint64(0)
        capi.check(self.L.stocs_score_best_device(self.h, dT, n, dL, id_offset, C.byref(key)))
        if key.value == 0:
            return 0.0, -1
        s = C.c_float(0); i = C.c_uint32(0)
        self.L.stocs_unpack_best(key.value, C.byref(s), C.byref(i))
        return s.value, int(i.value)

    def set_stream(self, hip_stream):
        capi.check(self.L.stocs_set_stream(self.h, C.c_void_p(hip_stream)))

    def score_best_device_async(self, dT, n, dL, id_offset, d_key8):
        """Scores + arg-max key in ONE launch (the arg-max is the scoring kernel's epilogue), nothing synchronised."""
        capi.check(self.L.stocs_score_best_device_async(self.h, dT, n, dL, id_offset, C.c_void_p(d_key8)))

    def best_device_async(self, dL, n, id_offset, d_key8):
        capi.check(self.L.stocs_best_device_async(self.h, dL, n, id_offset, C.c_void_p(d_key8)))

    def sync(self):
        capi.check(self.L.stocs_sync(self.h))

    def last_call_timing(self, which):
        """[(step, milliseconds)] of the last find_congruent_all (0), make_transforms (1) or compute_best_transform (2):
        host wall clock between the call's own synchronisation points, always recorded by the library."""
        labels = (C.c_char_p * 18)()   # which = 3: the phases of the last run_trials
        ms = (C.c_double * 18)()
        n = C.c_int(0)
        capi.check(self.L.stocs_last_call_timing(self.h, which, labels, ms, 18, C.byref(n)))
        return [(labels[i].decode(), float(ms[i])) for i in range(n.value)]

    def time_score_kernel(self, dT, n, dL, reps):
        ms = C.c_float(0)
        capi.check(self.L.stocs_time_score_kernel(self.h, dT, n, dL, reps, C.byref(ms)))
        return ms.value


def cluster_poses(poses16, lcp, acceptable_fraction, best_score, maximum_pose_count, min_distance, min_angle, sym):
    L = capi.load()
    p, pp = capi.f32(poses16); l, pl = capi.f32(lcp); s, ps = capi.f32(sym)
    out = np.zeros(max(len(l), 1), np.int32)
    n = C.c_int(0)
    capi.check(L.stocs_cluster_poses(pp, pl, len(l), acceptable_fraction, best_score, maximum_pose_count, min_distance,
                                     min_angle, ps, out.ctypes.data_as(capi._ip), len(out), C.byref(n)))
    return out[:n.value]


def ingest_scene(depth_u16, prob_u16, K, depth_scale, voxel_size=0.005, class_threshold=0.10, device=-1, normal_method=0):
    """GPU scene ingest (stocs_ingest_scene): returns pos, nrm, prob, pixel arrays.  normal_method 0: depth-gradient normals (the
    published LINEMOD method, what the reference asks OpenCV for), 1: the 5x5 plane fit of rounds 1-2 (the golden fixtures)."""
    L = capi.load()
    d = np.ascontiguousarray(depth_u16, np.uint16); p = np.ascontiguousarray(prob_u16, np.uint16)
    H, W = d.shape
    cam = capi.Camera(K[0], K[1], K[2], K[3], depth_scale, W, H, normal_method)
    cap = W * H
    pos = np.zeros((cap, 3), np.float32); nrm = np.zeros((cap, 3), np.float32); pr = np.zeros(cap, np.float32); px = np.zeros((cap, 2), np.int32)
    n = C.c_int(0)
    capi.check(L.stocs_ingest_scene(C.byref(cam), d.ctypes.data_as(C.POINTER(C.c_uint16)), p.ctypes.data_as(C.POINTER(C.c_uint16)), voxel_size,
                                    class_threshold, device, pos.ctypes.data_as(capi._fp), nrm.ctypes.data_as(capi._fp), pr.ctypes.data_as(capi._fp),
                                    px.ctypes.data_as(capi._ip), cap, C.byref(n)))
    k = n.value
    return pos[:k].copy(), nrm[:k].copy(), pr[:k].copy(), px[:k].copy()


def preprocess_model(raw_pos, normal_radius, voxel_size, model_scale=1.0, device=-1):
    """GPU model preprocessing (stocs_preprocess_model): returns voxelised pos, unit nrm."""
    L = capi.load()
    raw, pr = capi.f32(raw_pos)
    cap = len(raw)
    pos = np.zeros((cap, 3), np.float32); nrm = np.zeros((cap, 3), np.float32)
    n = C.c_int(0)
    capi.check(L.stocs_preprocess_model(pr, len(raw), normal_radius, voxel_size, model_scale, device, pos.ctypes.data_as(capi._fp),
                                        nrm.ctypes.data_as(capi._fp), cap, C.byref(n)))
    return pos[:n.value].copy(), nrm[:n.value].copy()


def icp_point_to_plane(src_pos, tgt_pos, tgt_nrm, max_iterations=5, max_correspondence_distance=0.035, device=-1):
    """GPU point-to-plane ICP (stocs_icp_point_to_plane): returns (T 4x4 float32 source->target, n_correspondences)."""
    L = capi.load()
    s, ps = capi.f32(src_pos); t, pt = capi.f32(tgt_pos); n, pn = capi.f32(tgt_nrm)
    T = np.zeros(16, np.float32)
    nc = C.c_int(0)
    capi.check(L.stocs_icp_point_to_plane(ps, len(s), pt, pn, len(t), max_iterations, max_correspondence_distance, device,
                                          T.ctypes.data_as(capi._fp), C.byref(nc)))
    return T.reshape(4, 4).T.copy(), nc.value
