"""ctypes binding of libstocs_hip.so (include/stocs_hip.h).  Fails loudly when the library is
missing -- there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstocs_hip.so")

STOCS_OK = 0
ERR_NAMES = {0: "OK", -1: "INVALID", -2: "NO_DEVICE", -3: "HIP", -4: "CAPACITY", -5: "STATE", -6: "NOMEM"}


class StocsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("stocs_hip error %s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("distance_threshold", C.c_float), ("ppf_tr_discretization", C.c_int),
                ("ppf_rot_discretization", C.c_int), ("plane_threshold", C.c_float),
                ("min_distance_base", C.c_float), ("internal_angle_threshold", C.c_float),
                ("lcp_normal_angle", C.c_float), ("image_width", C.c_int), ("image_height", C.c_int),
                ("number_of_bases", C.c_int), ("maximum_congruent_sets", C.c_int)]


class TrialResult(C.Structure):
    _fields_ = [("n_bases", C.c_int32), ("n_candidates", C.c_int32), ("n_quads", C.c_int64), ("best_lcp", C.c_float), ("best_index", C.c_int32),
                ("best_pose16", C.c_float * 16)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("cx", C.c_float), ("fy", C.c_float), ("cy", C.c_float), ("depth_scale", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("normal_method", C.c_int)]


# name -> (restype, argtypes); every symbol declared in include/stocs_hip.h
_fp, _ip, _u8p, _vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_void_p
_i64p = C.POINTER(C.c_int64)
_intp = C.POINTER(C.c_int)
SIGNATURES = {
    "stocs_default_params": (None, [C.POINTER(Params)]),
    "stocs_last_error": (C.c_char_p, []),
    "stocs_version": (C.c_char_p, []),
    "stocs_ctx_create": (C.c_int, [C.POINTER(Params), _fp, _fp, _fp, _ip, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "stocs_ctx_destroy": (C.c_int, [_vp]),
    "stocs_ctx_set_scene": (C.c_int, [_vp, _fp, _fp, _fp, _ip, C.c_int]),
    "stocs_get_centroids": (C.c_int, [_vp, _fp, _fp]),
    "stocs_get_sizes": (C.c_int, [_vp, _intp, _intp]),
    "stocs_set_edge_map": (C.c_int, [_vp, _u8p]),
    "stocs_reset_trial": (C.c_int, [_vp]),
    "stocs_ppf_compute_host": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, _ip]),
    "stocs_index_exists": (C.c_int, [_vp, _ip, _intp]),
    "stocs_index_lookup": (C.c_int, [_vp, _ip, _ip, C.c_int64, _i64p]),
    "stocs_index_stats": (C.c_int, [_vp, _i64p, _i64p, _i64p]),
    "stocs_index_save": (C.c_int, [_vp, C.c_char_p]),
    "stocs_index_load": (C.c_int, [_vp, C.c_char_p]),
    "stocs_sample_bases": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_float, _ip, _fp, _ip]),
    "stocs_get_segment": (C.c_int, [_vp, _ip, C.c_int, _intp]),
    "stocs_get_scene": (C.c_int, [_vp, _fp, _fp, _fp, _ip]),
    "stocs_png_read": (C.c_int, [C.c_char_p, _intp, _intp, _intp, _intp, _vp, C.c_int64]),
    "stocs_ply_read": (C.c_int, [C.c_char_p, _fp, _fp, C.c_int, _intp, _intp]),
    "stocs_ply_write": (C.c_int, [C.c_char_p, _fp, _fp, C.c_int, C.c_float]),
    "stocs_set_bases": (C.c_int, [_vp, C.c_int, _ip, _fp]),
    "stocs_clear_bases": (C.c_int, [_vp]),
    "stocs_num_bases": (C.c_int, [_vp]),
    "stocs_class_pass": (C.c_int, [_vp, C.c_int, _ip, _fp, _fp]),
    "stocs_ppf_filter_check": (C.c_int, [_vp, C.c_uint64, C.c_int64, _i64p, _i64p, _i64p]),
    "stocs_weight_fix_check": (C.c_int, [_vp, _i64p]),
    "stocs_try_sampled_base": (C.c_int, [_vp, _ip, _fp, _intp]),
    "stocs_draw": (C.c_int, [_vp, _fp, C.c_int, C.c_uint64, _intp]),
    "stocs_find_congruent_all": (C.c_int, [_vp, _i64p]),
    "stocs_get_quads": (C.c_int, [_vp, C.c_int, _ip, C.c_int64, _i64p]),
    "stocs_get_quads_at": (C.c_int, [_vp, C.c_int, _i64p, C.c_int, _ip]),
    "stocs_cone_cells_host": (C.c_int, [_fp, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _intp, _intp]),
    "stocs_make_transforms": (C.c_int, [_vp, C.c_int, C.c_uint64, _intp]),
    "stocs_rigid_transform": (C.c_int, [_vp, _ip, _ip, _fp, _fp, _intp]),
    "stocs_get_candidates": (C.c_int, [_vp, _fp, _fp, _fp, _ip, C.c_int, _intp]),
    "stocs_score_transforms": (C.c_int, [_vp, _fp, C.c_int, _fp]),
    "stocs_score_transforms_device": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "stocs_score_best_device": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_uint32, C.POINTER(C.c_uint64)]),
    "stocs_lcp_detail": (C.c_int, [_vp, _fp, _ip, _u8p]),
    "stocs_lcp_hit_count": (C.c_int, [_vp, _vp, C.c_int, _i64p, _i64p]),
    "stocs_verify_all": (C.c_int, [_vp, _fp, _intp, _fp]),
    "stocs_run_trials": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(TrialResult)]),
    "stocs_trials_get_bases": (C.c_int, [_vp, C.c_int, _ip, _fp, _ip, C.c_int, _intp]),
    "stocs_trials_get_quad_counts": (C.c_int, [_vp, C.c_int, _i64p, C.c_int, _intp]),
    "stocs_trials_get_candidates": (C.c_int, [_vp, C.c_int, _fp, _fp, _fp, _ip, C.c_int, _intp]),
    "stocs_best_device": (C.c_int, [_vp, _vp, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]),
    "stocs_pack_best": (C.c_uint64, [C.c_float, C.c_uint32]),
    "stocs_unpack_best": (None, [C.c_uint64, _fp, C.POINTER(C.c_uint32)]),
    "stocs_comm_unique_id": (C.c_int, [_vp]),
    "stocs_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "stocs_comm_destroy": (C.c_int, [_vp]),
    "stocs_allreduce_best": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64), _fp, C.c_uint32]),
    "stocs_cluster_poses": (C.c_int, [_fp, _fp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, _fp, _ip, C.c_int, _intp]),
    "stocs_ingest_scene": (C.c_int, [C.POINTER(Camera), C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_float, C.c_float, C.c_int, _fp, _fp, _fp, _ip, C.c_int, _intp]),
    "stocs_preprocess_model": (C.c_int, [_fp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, _fp, _fp, C.c_int, _intp]),
    "stocs_trim": (C.c_int, []),
    "stocs_icp_point_to_plane": (C.c_int, [_fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_float, C.c_int, _fp, _intp]),
    "stocs_device_alloc_count": (C.c_int64, []),
    "stocs_debug_stream_audit_selftest": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "stocs_debug_streams_overlap": (C.c_int, [_vp]),
    "stocs_debug_sort_pairs": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _fp, C.POINTER(C.c_uint32), C.c_int]),
    "stocs_last_call_timing": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.c_int, _intp]),
    "stocs_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "stocs_get_cull_state": (C.c_int, [_vp, _fp, _ip, _intp, _fp, _fp, C.c_int64, _i64p]),
    "stocs_model_patch_order": (C.c_int, [_fp, C.c_int, _ip, _fp]),
    "stocs_set_stream": (C.c_int, [_vp, _vp]),
    "stocs_best_device_async": (C.c_int, [_vp, _vp, C.c_int, C.c_uint32, _vp]),
    "stocs_score_best_device_async": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_uint32, _vp]),
    "stocs_sync": (C.c_int, [_vp]),
    "stocs_stream": (_vp, [_vp]),
    "stocs_time_score_kernel": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _fp]),
    "stocs_dev_alloc": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "stocs_dev_free": (C.c_int, [_vp, _vp]),
    "stocs_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "stocs_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
}

_LIB = None


def load():
    """Load libstocs_hip.so; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C model_matching_amd/csrc).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != STOCS_OK:
        raise StocsError(rc, load().stocs_last_error().decode())


def default_params(**kw) -> Params:
    p = Params()
    load().stocs_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)
