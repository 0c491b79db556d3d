#!/usr/bin/env python3
"""Derives flat cloud fixtures from the reference's example DATA files (PNG images and raw PLY
vertex lists under /root/reference/examples and /root/reference/models) and writes them to
tests/golden/example_<name>.npz.  Only data is read; no reference code is imported, compiled or run.

This is this repo's own restatement (numpy / PIL / scipy) of the two steps that sit UPSTREAM of the
hot path and are out of its scope:
  * scene ingest  -- rgbd::load_rgbd_data_sampled, reference src/rgbd.cpp:179-281: back-projection,
    PCL VoxelGrid (centroid per leaf), PCL RadiusOutlierRemoval (> 10 points within 2*voxel+5 mm),
    z in (0,2], re-projection to the pixel, class-probability threshold, surface normals;
  * model preprocessing -- stocs::pre_process_model, reference src/stocs.cpp:28-60: normals from
    radius neighbourhoods, flipped to point away from the model origin, PCL VoxelGrid.
The arithmetic of PCL / OpenCV-contrib (LINEMOD normals) is not available here, so agreement with the
clouds the reference would build is PARITY-UNPINNED; hot-path parity is defined GIVEN these clouds.
Parameters per data set: reference README.md:36-66 and src/stocs_match_one_object.cpp:4-24.

Run:  python tests/golden/make_example_fixtures.py
"""
import os
import sys

import numpy as np
from PIL import Image
from scipy.spatial import cKDTree

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

CONFIGS = {
    # name: scene dir, object, intrinsics (fx, cx, fy, cy), depth scale, model voxel, normal radius, model scale
    "ycb_024_bowl": dict(scene="examples/ycb", obj="024_bowl", K=(1066.778, 312.986, 1067.487, 241.310), ds=1 / 10000.0,
                         mvox=0.01, nrad=0.005, mscale=1.0),
    "linemod_obj_06": dict(scene="examples/linemod", obj="obj_06", K=(572.4114, 325.2611, 573.57043, 242.04899), ds=1 / 1000.0,
                           mvox=10.0, nrad=5.0, mscale=1.0 / 1000),
    "packed_dove": dict(scene="examples/packed", obj="dove", K=(615.957763671875, 308.1098937988281, 615.9578247070312, 246.33352661132812),
                        ds=1 / 8000.0, mvox=0.005, nrad=0.005, mscale=1.0),
}
VOXEL = 0.005          # stocs_match_one_object.cpp:7
CLASS_THRESHOLD = 0.10  # :12


def voxel_grid(points, leaf, extra=None):
    """PCL VoxelGrid: centroid of the points of each leaf, leaves visited in ascending linear index
    (x fastest).  extra (n,k) fields are averaged too (downsample_all_data)."""
    inv = 1.0 / leaf
    ijk = np.floor(points.astype(np.float64) * inv).astype(np.int64)
    ijk -= ijk.min(axis=0)
    dims = ijk.max(axis=0) + 1
    lin = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    order = np.argsort(lin, kind="stable")
    lin_s = lin[order]
    starts = np.flatnonzero(np.r_[True, lin_s[1:] != lin_s[:-1]])
    counts = np.diff(np.r_[starts, len(lin_s)])
    cen = np.add.reduceat(points[order].astype(np.float64), starts, axis=0) / counts[:, None]
    if extra is None:
        return cen.astype(np.float32)
    ext = np.add.reduceat(extra[order].astype(np.float64), starts, axis=0) / counts[:, None]
    return cen.astype(np.float32), ext.astype(np.float32)


def depth_normals(P, valid, win=5):
    """Per-pixel normals from the back-projected point map: least-squares plane over a win x win
    window (stand-in for cv::rgbd::RgbdNormals LINEMOD with window 5, rgbd.cpp:203), oriented toward
    the camera; NaN where fewer than 6 valid pixels or at depth discontinuities."""
    H, W, _ = P.shape
    r = win // 2
    v = valid.astype(np.float64)
    def box(a):
        c = np.cumsum(np.cumsum(np.pad(a, ((r + 1, r), (r + 1, r))), axis=0), axis=1)
        return c[win:, win:] - c[:-win, win:] - c[win:, :-win] + c[:-win, :-win]
    X, Y, Z = (P[..., k].astype(np.float64) * v for k in range(3))
    n = box(v)
    sx, sy, sz = box(X), box(Y), box(Z)
    sxx, sxy, sxz, syy, syz, szz = box(X * X), box(X * Y), box(X * Z), box(Y * Y), box(Y * Z), box(Z * Z)
    nn = np.maximum(n, 1)
    mx, my, mz = sx / nn, sy / nn, sz / nn
    C = np.empty((H, W, 3, 3))
    C[..., 0, 0] = sxx / nn - mx * mx; C[..., 0, 1] = sxy / nn - mx * my; C[..., 0, 2] = sxz / nn - mx * mz
    C[..., 1, 1] = syy / nn - my * my; C[..., 1, 2] = syz / nn - my * mz; C[..., 2, 2] = szz / nn - mz * mz
    C[..., 1, 0] = C[..., 0, 1]; C[..., 2, 0] = C[..., 0, 2]; C[..., 2, 1] = C[..., 1, 2]
    w, V = np.linalg.eigh(C)
    nrm = V[..., :, 0]                                  # smallest eigenvalue
    flip = (nrm * P).sum(-1) > 0                        # toward the camera: n . p < 0
    nrm[flip] *= -1
    # depth discontinuity: local z range
    zr = np.where(valid, P[..., 2], np.nan)
    bad = (n < 6) | ~valid | (w[..., 0] > 1e-5)         # plane residual variance > (3 mm)^2
    nrm[bad] = np.nan
    return nrm.astype(np.float32)


def ingest_scene(cfg):
    fx, cx, fy, cy = cfg["K"]
    d = np.array(Image.open(os.path.join(REF, cfg["scene"], "depth.png"))).astype(np.float32) * np.float32(cfg["ds"])
    prob = np.array(Image.open(os.path.join(REF, cfg["scene"], "probability_maps", cfg["obj"] + ".png")))
    H, W = d.shape
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    P = np.stack([((jj - cx) * d / fx), ((ii - cy) * d / fy), d], axis=-1).astype(np.float32)   # rgbd.cpp:214-216
    normals = depth_normals(P, d > 0)
    cloud = voxel_grid(P.reshape(-1, 3), VOXEL)                                                  # :228-231
    tree = cKDTree(cloud.astype(np.float64))
    k = tree.query_ball_point(cloud.astype(np.float64), 2 * VOXEL + 0.005, return_length=True)   # :233-237
    cloud = cloud[k > 10]
    pos, nrm, pr, pix = [], [], [], []
    for pt in cloud:
        if not np.isfinite(pt[2]) or pt[2] <= 0 or pt[2] > 2.0:                                  # :243-244
            continue
        col = int((np.float32(fx) * pt[0] + np.float32(cx) * pt[2]) / pt[2])                     # :251-253
        row = int((np.float32(fy) * pt[1] + np.float32(cy) * pt[2]) / pt[2])
        if not (0 <= row < H and 0 <= col < W):
            continue
        cp = np.float32(float(prob[row, col]) * (1.0 / 10000))                                   # :255
        if cp < CLASS_THRESHOLD:
            continue
        n = normals[row, col]
        if not np.isfinite(n).all() or (n == 0).all():                                           # :264-267
            continue
        pos.append(pt); nrm.append(n); pr.append(cp); pix.append((row, col))
    return (np.array(pos, np.float32), np.array(nrm, np.float32), np.array(pr, np.float32), np.array(pix, np.int32))


def read_ply_xyz(path):
    with open(path) as f:
        n = 0
        for line in f:
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line.startswith("end_header"):
                break
        return np.loadtxt(f, max_rows=n, dtype=np.float64)[:, :3]


def preprocess_model(cfg):
    pts = read_ply_xyz(os.path.join(REF, "models", cfg["obj"], "textured_vertices.ply"))
    tree = cKDTree(pts)
    nb = tree.query_ball_point(pts, cfg["nrad"])
    nrm = np.full(pts.shape, np.nan)
    for i, idx in enumerate(nb):                        # pcl::NormalEstimation with radius search (rgbd.cpp:72-83)
        if len(idx) < 3:
            continue
        q = pts[idx] - pts[idx].mean(axis=0)
        w, V = np.linalg.eigh(q.T @ q)
        n = V[:, 0]
        if np.dot(n, -pts[i]) < 0:                      # flipNormalTowardsViewpoint(0,0,0)
            n = -n
        nrm[i] = -n                                     # stocs.cpp:47-52 negation -> away from the origin
    ok = np.isfinite(nrm).all(axis=1)
    cen, navg = voxel_grid(pts[ok].astype(np.float32), cfg["mvox"], nrm[ok].astype(np.float32))  # stocs.cpp:54-57
    fin = np.isfinite(navg).all(axis=1) & (np.linalg.norm(navg, axis=1) > 0)
    cen, navg = cen[fin], navg[fin]
    navg = navg / np.linalg.norm(navg, axis=1, keepdims=True)                                   # set_normal, point3d.hpp:43-45
    return (cen * np.float32(cfg["mscale"])).astype(np.float32), navg.astype(np.float32)


def main():
    for name, cfg in CONFIGS.items():
        spos, snrm, sprob, spix = ingest_scene(cfg)
        mpos, mnrm = preprocess_model(cfg)
        out = dict(scene_pos=spos, scene_nrm=snrm, scene_prob=sprob, scene_pixel=spix, model_pos=mpos, model_nrm=mnrm)
        ep = os.path.join(REF, cfg["scene"], "probability_maps", "edge.png")
        if os.path.exists(ep):
            out["edge_map"] = np.array(Image.open(ep)).astype(np.uint8)
        path = os.path.join(OUT, "example_%s.npz" % name)
        np.savez_compressed(path, **out)
        print(name, "|S| =", len(spos), "|M| =", len(mpos), "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    sys.exit(main())
