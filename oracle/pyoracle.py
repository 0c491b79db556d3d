"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    _fields_ = [("distance_threshold", C.c_float), ("ppf_tr_discretization", C.c_int),
                ("ppf_rot_discretization", C.c_int), ("plane_threshold", C.c_float),
                ("min_distance_base", C.c_float), ("internal_angle_threshold", C.c_float),
                ("image_width", C.c_int), ("image_height", C.c_int)]


class RunResult(C.Structure):
    _fields_ = [("n_bases", C.c_int), ("n_quads_total", C.c_int), ("n_candidates", C.c_int),
                ("best_lcp", C.c_float), ("best_index", C.c_int), ("best_pose16", C.c_float * 16),
                ("t_sample_s", C.c_double), ("t_congruent_s", C.c_double), ("t_verify_s", C.c_double)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "stocs_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp, ip, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        vp = C.c_void_p
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_ppf_closest_bin.argtypes = [C.c_int, C.c_int]
        L.orc_ppf_closest_bin.restype = C.c_int
        L.orc_ppf_compute.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, ip]
        L.orc_index_build.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int]
        L.orc_index_build.restype = vp
        L.orc_set_index_build_threads.argtypes = [C.c_int]
        L.orc_set_index_build_threads.restype = None
        L.orc_index_free.argtypes = [vp]
        L.orc_index_lookup.argtypes = [vp, ip, ip, C.c_int64]
        L.orc_index_lookup.restype = C.c_int64
        L.orc_index_exists.argtypes = [vp, ip]
        L.orc_index_exists.restype = C.c_int
        L.orc_index_num_pairs.argtypes = [vp]
        L.orc_index_num_pairs.restype = C.c_int64
        L.orc_index_lit_build.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int]
        L.orc_index_lit_build.restype = vp
        L.orc_index_lit_free.argtypes = [vp]
        L.orc_index_lit_lookup.argtypes = [vp, ip, ip, C.c_int64]
        L.orc_index_lit_lookup.restype = C.c_int64
        L.orc_index_lit_num_keys.argtypes = [vp]
        L.orc_index_lit_num_keys.restype = C.c_int64
        L.orc_ctx_create.argtypes = [C.POINTER(Params), fp, fp, fp, ip, C.c_int, fp, fp, C.c_int, C.c_int]
        L.orc_ctx_create.restype = vp
        L.orc_ctx_destroy.argtypes = [vp]
        L.orc_get_centroids.argtypes = [vp, fp, fp]
        L.orc_get_scene.argtypes = [vp, fp, fp, fp]
        L.orc_get_model.argtypes = [vp, fp]
        L.orc_set_edge_map.argtypes = [vp, u8p]
        L.orc_ctx_index.argtypes = [vp]
        L.orc_ctx_index.restype = vp
        L.orc_rng.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_rng.restype = C.c_uint64
        L.orc_draw.argtypes = [fp, C.c_int, C.c_uint64]
        L.orc_draw.restype = C.c_int
        L.orc_sample_class_base.argtypes = [vp, C.c_uint64, C.c_uint64, ip, fp]
        L.orc_sample_class_base.restype = C.c_int
        L.orc_sample_instance_base.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_float, C.c_int, ip, fp]
        L.orc_sample_instance_base.restype = C.c_int
        L.orc_get_segment.argtypes = [vp, ip, C.c_int]
        L.orc_get_segment.restype = C.c_int
        L.orc_class_pass.argtypes = [vp, C.c_int, ip, fp, fp]
        L.orc_segment_distance_and_invariants.argtypes = [fp, fp, fp, fp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_segment_distance_and_invariants.restype = C.c_double
        L.orc_try_sampled_base.argtypes = [vp, ip, fp]
        L.orc_try_sampled_base.restype = C.c_int
        L.orc_find_congruent.argtypes = [vp, ip, C.c_float, C.c_float, ip, C.c_int64]
        L.orc_find_congruent.restype = C.c_int64
        L.orc_find_congruent_seq.argtypes = [vp, ip, C.c_float, C.c_float, ip, C.c_int64]
        L.orc_find_congruent_seq.restype = C.c_int64
        L.orc_normalset_params.argtypes = [C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int), fp]
        L.orc_cone_samples.argtypes = [C.c_float]
        L.orc_cone_samples.restype = C.c_int
        L.orc_index_normal.argtypes = [fp]
        L.orc_index_normal.restype = C.c_int
        L.orc_model_ratio.argtypes = [vp, fp]
        L.orc_model_ratio.restype = C.c_float
        L.orc_rigid_transform.argtypes = [vp, ip, ip, fp, fp]
        L.orc_rigid_transform.restype = C.c_int
        L.orc_nn.argtypes = [vp, fp, C.c_float]
        L.orc_nn.restype = C.c_int
        L.orc_nn_brute.argtypes = [vp, fp, C.c_float, C.POINTER(C.c_int)]
        L.orc_nn_brute.restype = C.c_int
        L.orc_lcp.argtypes = [vp, fp]
        L.orc_lcp.restype = C.c_float
        L.orc_lcp_batch.argtypes = [vp, fp, C.c_int, fp, C.c_int]
        L.orc_lcp_detail.argtypes = [vp, fp, ip, u8p]
        L.orc_lcp_batch_exact.argtypes = [vp, fp, C.c_int, fp, C.POINTER(C.c_double), C.c_int]
        L.orc_best.argtypes = [fp, C.c_int, fp]
        L.orc_best.restype = C.c_int
        L.orc_normal_compatible.argtypes = [C.c_float]
        L.orc_normal_compatible.restype = C.c_int
        L.orc_internal_angle_reject.argtypes = [C.c_float, C.c_float]
        L.orc_internal_angle_reject.restype = C.c_int
        L.orc_greedy_clustering.argtypes = [fp, fp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float,
                                            C.c_float, fp, ip, C.c_int]
        L.orc_greedy_clustering.restype = C.c_int
        L.orc_pose_diff.argtypes = [fp, fp, fp, fp, fp]
        L.orc_run.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(RunResult)]
        L.orc_run.restype = C.c_int
        L.orc_run_mode.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(RunResult)]
        L.orc_run_mode.restype = C.c_int
        L.orc_get_candidates.argtypes = [vp, fp, fp, ip, C.c_int]
        L.orc_get_candidates.restype = C.c_int
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def default_params(**kw) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ppf_compute(p1, n1, p2, n2, tr=5, rot=5, mode=0):
    out = np.zeros(4, np.int32)
    a, pa = _f(p1); b, pb = _f(n1); c, pc = _f(p2); d, pd = _f(n2)
    lib().orc_ppf_compute(pa, pb, pc, pd, tr, rot, mode, out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def closest_bin(v, d):
    return lib().orc_ppf_closest_bin(int(v), int(d))


def normalize_rows(n):
    """Point3D::set_normal (point3d.hpp:43-45) in float32 with the oracle's operation order."""
    n = np.asarray(n, np.float32)
    z = n[:, 0] * n[:, 0] + (n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2])
    s = np.sqrt(z, dtype=np.float32)
    out = n.copy()
    ok = z > 0
    out[ok] = n[ok] / s[ok, None]
    return out


class Index:
    def __init__(self, pos, nrm, tr=5, rot=5, literal=False):
        self.pos, pp = _f(pos)
        self.nrm, pn = _f(nrm)
        self.literal = literal
        L = lib()
        if literal:
            self.h = L.orc_index_lit_build(pp, pn, len(self.pos), tr, rot)
        else:
            self.h = L.orc_index_build(pp, pn, len(self.pos), tr, rot)

    def lookup(self, key):
        k, pk = _i(key)
        L = lib()
        fn = L.orc_index_lit_lookup if self.literal else L.orc_index_lookup
        n = fn(self.h, pk, None, 0)
        out = np.zeros((n, 2), np.int32)
        if n:
            fn(self.h, pk, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def exists(self, key):
        k, pk = _i(key)
        if self.literal:
            return lib().orc_index_lit_lookup(self.h, pk, None, 0) > 0
        return bool(lib().orc_index_exists(self.h, pk))

    def num_pairs(self):
        return lib().orc_index_num_pairs(self.h)

    def num_keys(self):
        return lib().orc_index_lit_num_keys(self.h)

    def __del__(self):
        try:
            if self.h:
                (lib().orc_index_lit_free if self.literal else lib().orc_index_free)(self.h)
                self.h = None
        except Exception:
            pass


class Oracle:
    """Mirror of stocs::stocs_estimator (include/stocs.hpp:16-180) over flat clouds."""

    def __init__(self, scene_pos, scene_nrm, scene_prob, scene_pixel, model_pos, model_nrm,
                 params: Params | None = None, build_index=True):
        L = lib()
        self.prm = params or default_params()
        self.sp, psp = _f(scene_pos)
        self.sn, psn = _f(scene_nrm)
        self.spr, pspr = _f(scene_prob)
        if scene_pixel is None:
            scene_pixel = np.zeros((len(self.sp), 2), np.int32)
        self.spx, pspx = _i(scene_pixel)
        self.mp, pmp = _f(model_pos)
        self.mn, pmn = _f(model_nrm)
        self.nS, self.nM = len(self.sp), len(self.mp)
        self.h = L.orc_ctx_create(C.byref(self.prm), psp, psn, pspr, pspx, self.nS, pmp, pmn, self.nM,
                                  1 if build_index else 0)

    def __del__(self):
        try:
            if self.h:
                lib().orc_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def centroids(self):
        s, ps = _f(np.zeros(3)); m, pm = _f(np.zeros(3))
        lib().orc_get_centroids(self.h, ps, pm)
        return s, m

    def scene_centred(self):
        pos = np.zeros((self.nS, 3), np.float32)
        lib().orc_get_scene(self.h, pos.ctypes.data_as(C.POINTER(C.c_float)), None, None)
        return pos

    def scene_class_prob(self):
        cp = np.zeros(self.nS, np.float32)
        lib().orc_get_scene(self.h, None, None, cp.ctypes.data_as(C.POINTER(C.c_float)))
        return cp

    def model_centred(self):
        pos = np.zeros((self.nM, 3), np.float32)
        lib().orc_get_model(self.h, pos.ctypes.data_as(C.POINTER(C.c_float)))
        return pos

    def set_edge_map(self, edge):
        e = np.ascontiguousarray(edge, np.uint8)
        lib().orc_set_edge_map(self.h, e.ctypes.data_as(C.POINTER(C.c_uint8)))

    def index_lookup(self, key):
        k, pk = _i(key)
        ix = lib().orc_ctx_index(self.h)
        n = lib().orc_index_lookup(ix, pk, None, 0)
        out = np.zeros((n, 2), np.int32)
        if n:
            lib().orc_index_lookup(ix, pk, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def index_exists(self, key):
        k, pk = _i(key)
        return bool(lib().orc_index_exists(lib().orc_ctx_index(self.h), pk))

    def sample_class_base(self, seed, attempt):
        ids = np.zeros(4, np.int32); inv = np.zeros(2, np.float32)
        ok = lib().orc_sample_class_base(self.h, seed, attempt, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                         inv.ctypes.data_as(C.POINTER(C.c_float)))
        return bool(ok), ids, inv

    def sample_instance_base(self, seed, attempt, dispersion, base_num):
        ids = np.zeros(4, np.int32); inv = np.zeros(2, np.float32)
        ok = lib().orc_sample_instance_base(self.h, seed, attempt, dispersion, base_num,
                                            ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                            inv.ctypes.data_as(C.POINTER(C.c_float)))
        return bool(ok), ids, inv

    def get_segment(self):
        n = lib().orc_get_segment(self.h, None, 0)
        out = np.zeros(n, np.int32)
        if n:
            lib().orc_get_segment(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def class_pass(self, k, b3, w_in):
        b, pb = _i(b3)
        w, pw = _f(w_in)
        out = np.zeros_like(w)
        lib().orc_class_pass(self.h, k, pb, pw, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def try_sampled_base(self, ids):
        ids, pi = _i(np.array(ids).copy())
        inv = np.zeros(2, np.float32)
        ok = lib().orc_try_sampled_base(self.h, pi, inv.ctypes.data_as(C.POINTER(C.c_float)))
        return bool(ok), ids, inv

    def find_congruent(self, ids, inv1, inv2):
        ids, pi = _i(ids)
        n = lib().orc_find_congruent(self.h, pi, inv1, inv2, None, 0)
        out = np.zeros((n, 4), np.int32)
        if n:
            lib().orc_find_congruent(self.h, pi, inv1, inv2, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def find_congruent_seq(self, ids, inv1, inv2):
        ids, pi = _i(ids)
        n = lib().orc_find_congruent_seq(self.h, pi, inv1, inv2, None, 0)
        out = np.zeros((n, 4), np.int32)
        if n:
            lib().orc_find_congruent_seq(self.h, pi, inv1, inv2, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def rigid_transform(self, ids, quad):
        ids, pi = _i(ids); q, pq = _i(quad)
        T = np.zeros(16, np.float32); P = np.zeros(16, np.float32)
        ok = lib().orc_rigid_transform(self.h, pi, pq, T.ctypes.data_as(C.POINTER(C.c_float)),
                                       P.ctypes.data_as(C.POINTER(C.c_float)))
        return bool(ok), T, P

    def nn(self, q, sqdist):
        q, pq = _f(q)
        return lib().orc_nn(self.h, pq, sqdist)

    def nn_brute(self, q, sqdist):
        q, pq = _f(q)
        t = C.c_int(0)
        r = lib().orc_nn_brute(self.h, pq, sqdist, C.byref(t))
        return r, t.value

    def lcp(self, T16):
        T, pT = _f(T16)
        return lib().orc_lcp(self.h, pT)

    def lcp_batch(self, T16, nthreads=1):
        T, pT = _f(T16)
        n = T.size // 16
        out = np.zeros(n, np.float32)
        lib().orc_lcp_batch(self.h, pT, n, out.ctypes.data_as(C.POINTER(C.c_float)), nthreads)
        return out

    def lcp_batch_exact(self, T16, nthreads=1):
        """(reference float-accumulated scores, the same matches summed in double)"""
        T, pT = _f(T16)
        n = T.size // 16
        out = np.zeros(n, np.float32); ex = np.zeros(n, np.float64)
        lib().orc_lcp_batch_exact(self.h, pT, n, out.ctypes.data_as(C.POINTER(C.c_float)), ex.ctypes.data_as(C.POINTER(C.c_double)), nthreads)
        return out, ex

    def lcp_detail(self, T16):
        T, pT = _f(T16)
        hit = np.zeros(self.nM, np.int32); counted = np.zeros(self.nM, np.uint8)
        lib().orc_lcp_detail(self.h, pT, hit.ctypes.data_as(C.POINTER(C.c_int32)),
                             counted.ctypes.data_as(C.POINTER(C.c_uint8)))
        return hit, counted

    def run(self, seed, number_of_bases=100, maximum_congruent_sets=200, instance_mode=False, dispersion=0.9):
        r = RunResult()
        lib().orc_run_mode(self.h, seed, number_of_bases, maximum_congruent_sets, 1 if instance_mode else 0, dispersion, C.byref(r))
        return r

    def candidates(self):
        n = lib().orc_get_candidates(self.h, None, None, None, 0)
        T = np.zeros((n, 16), np.float32); P = np.zeros((n, 16), np.float32); b = np.zeros(n, np.int32)
        if n:
            lib().orc_get_candidates(self.h, T.ctypes.data_as(C.POINTER(C.c_float)),
                                     P.ctypes.data_as(C.POINTER(C.c_float)),
                                     b.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return T, P, b


def best(lcp):
    l, pl = _f(lcp)
    s = C.c_float(0)
    i = lib().orc_best(pl, len(l), C.byref(s))
    return i, s.value


def greedy_clustering(poses16, lcp, acceptable_fraction, best_score, maximum_pose_count, min_distance,
                      min_angle, sym):
    p, pp = _f(poses16); l, pl = _f(lcp); s, ps = _f(sym)
    out = np.zeros(len(l), np.int32)
    n = lib().orc_greedy_clustering(pp, pl, len(l), acceptable_fraction, best_score, maximum_pose_count,
                                    min_distance, min_angle, ps, out.ctypes.data_as(C.POINTER(C.c_int32)),
                                    len(l))
    return out[:n]


def pose_diff(test16, base16, sym):
    a, pa = _f(test16); b, pb = _f(base16); s, ps = _f(sym)
    r = C.c_float(0); t = C.c_float(0)
    lib().orc_pose_diff(pa, pb, ps, C.byref(r), C.byref(t))
    return r.value, t.value


def set_index_build_threads(n):
    """Threads for the feature evaluation of the oracle's index build (the index itself does not depend on the count)."""
    lib().orc_set_index_build_threads(int(n))
