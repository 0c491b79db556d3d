#!/usr/bin/env python3
"""How well does a recovered pose explain the reference's OWN input images?

The reference commits no expected pose for its three example frames: the only reference-held evidence a winner can be
checked against is the data the frame itself carries -- the 16-bit depth image and the class-probability map under
examples/<frame>/ (held here as tests/golden/example_*_raw.npz).  The check inverts the back-projection of the scene ingest
(reference src/rgbd.cpp:179-281: z = depth * scale, x = (col - cx) z / fx, y = (row - cy) z / fy): the model cloud is moved through
the pose, its camera-facing points (transformed normal towards the camera) are projected through K into the 640x480 image, and at
every such pixel the model's depth is compared with the observed one and the class probability is read off the map.

    depth_agreement(...) -> {"visible_points", "in_image", "with_depth", "within_5mm", "within_10mm", "median_abs_dz_mm",
                             "on_mask", "mean_class_prob"}    (fractions of the visible points)

Used by tests/test_driver_gpu.py (assertions with generous floors) and tools/frame_latency.py (reported numbers).  numpy only."""
import numpy as np


def depth_agreement(pose, model_pos, model_nrm, depth_u16, prob_u16, K, depth_scale, class_threshold=0.10):
    """pose: 4x4 (or 3x4) model -> camera, metres.  model_pos / model_nrm: the model cloud the pose was estimated with (n, 3).
    K = (fx, cx, fy, cy) as the reference's driver orders them (stocs_match_one_object.cpp:20)."""
    P = np.asarray(pose, np.float64).reshape(-1, 4)[:3]
    R, t = P[:, :3], P[:, 3]
    fx, cx, fy, cy = (float(v) for v in K)
    pts = np.asarray(model_pos, np.float64) @ R.T + t
    nrm = np.asarray(model_nrm, np.float64) @ R.T
    facing = (nrm * pts).sum(1) < 0.0                      # the surface normal points back at the camera (origin)
    pts = pts[facing & (pts[:, 2] > 1e-6)]
    n_vis = int(len(pts))
    out = {"visible_points": n_vis, "in_image": 0.0, "with_depth": 0.0, "within_5mm": 0.0, "within_10mm": 0.0, "median_abs_dz_mm": None,
           "on_mask": 0.0, "mean_class_prob": 0.0}
    if n_vis == 0:
        return out
    H, W = depth_u16.shape
    col = np.floor(fx * pts[:, 0] / pts[:, 2] + cx + 0.5).astype(np.int64)
    row = np.floor(fy * pts[:, 1] / pts[:, 2] + cy + 0.5).astype(np.int64)
    inside = (row >= 0) & (row < H) & (col >= 0) & (col < W)
    out["in_image"] = float(inside.mean())
    if not inside.any():
        return out
    r, c, z = row[inside], col[inside], pts[inside, 2]
    zo = depth_u16[r, c].astype(np.float64) * float(depth_scale)
    have = zo > 0
    dz = np.abs(zo[have] - z[have])
    out["with_depth"] = float(have.sum() / n_vis)
    out["within_5mm"] = float((dz <= 0.005).sum() / n_vis)
    out["within_10mm"] = float((dz <= 0.010).sum() / n_vis)
    out["median_abs_dz_mm"] = float(np.median(dz) * 1e3) if len(dz) else None
    cp = prob_u16[r, c].astype(np.float64) * (1.0 / 10000.0)       # rgbd.cpp:255
    out["on_mask"] = float((cp >= class_threshold).sum() / n_vis)
    out["mean_class_prob"] = float(cp.sum() / n_vis)
    return out


def pose_matrix_from_colmajor16(p16):
    return np.asarray(p16, np.float64).reshape(4, 4).T
