"""numpy restatement of the two steps UPSTREAM of the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

  * scene ingest        rgbd::load_rgbd_data_sampled, reference src/rgbd.cpp:179-281
  * model preprocessing stocs::pre_process_model (cloud part), reference src/stocs.cpp:28-60

The reference delegates the arithmetic to PCL (VoxelGrid, RadiusOutlierRemoval, NormalEstimation) and
OpenCV-contrib (RgbdNormals, LINEMOD method); neither library is available, so agreement with the
clouds the reference would build is PARITY-UNPINNED.  This restatement pins the GPU implementation
(model_matching_amd/csrc/ingest.hip) and generates tests/golden/example_*.npz.
Only tests/ and tests/golden/*.py may import it."""
import numpy as np
from scipy.spatial import cKDTree


def voxel_grid(points, leaf, extra=None):
    """pcl::VoxelGrid: centroid of the points of each leaf, leaves in ascending linear index (x fastest).
    Leaf coordinates in float32 as PCL computes them: floor(p * (1/leaf)).  extra fields are averaged."""
    pts = np.asarray(points, np.float32)
    inv = np.float64(1.0) / np.float64(np.float32(leaf))       # the GPU uses the same double reciprocal
    ijk = np.floor(pts.astype(np.float64) * inv).astype(np.int64)
    ijk -= ijk.min(axis=0)
    dims = ijk.max(axis=0) + 1
    lin = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    order = np.argsort(lin, kind="stable")
    lin_s = lin[order]
    starts = np.flatnonzero(np.r_[True, lin_s[1:] != lin_s[:-1]])
    counts = np.diff(np.r_[starts, len(lin_s)])
    cen = np.add.reduceat(pts[order].astype(np.float64), starts, axis=0) / counts[:, None]
    if extra is None:
        return cen.astype(np.float32)
    ext = np.add.reduceat(np.asarray(extra, np.float32)[order].astype(np.float64), starts, axis=0) / counts[:, None]
    return cen.astype(np.float32), ext.astype(np.float32)


def depth_normals(P, valid, win=5):
    """Least-squares plane over the valid pixels of a win x win window (normal_method 1: the stand-in of rounds 1-2 for
    RgbdNormals LINEMOD, rgbd.cpp:203; the golden fixtures hold its clouds), oriented toward the camera; NaN where unreliable."""
    H, W, _ = P.shape
    r = win // 2
    acc = {k: np.zeros((H, W)) for k in ("n", "x", "y", "z", "xx", "xy", "xz", "yy", "yz", "zz")}
    X, Y, Z = (P[..., k].astype(np.float64) for k in range(3))
    v = valid.astype(np.float64)
    # direct window sums in row-major (di, dj) order -- the order the GPU kernel uses
    for di in range(-r, r + 1):
        for dj in range(-r, r + 1):
            sl_dst = (slice(max(0, -di), H - max(0, di)), slice(max(0, -dj), W - max(0, dj)))
            sl_src = (slice(max(0, di), H - max(0, -di)), slice(max(0, dj), W - max(0, -dj)))
            vv = v[sl_src]
            x, y, z = X[sl_src] * vv, Y[sl_src] * vv, Z[sl_src] * vv
            acc["n"][sl_dst] += vv
            acc["x"][sl_dst] += x; acc["y"][sl_dst] += y; acc["z"][sl_dst] += z
            acc["xx"][sl_dst] += x * X[sl_src]; acc["xy"][sl_dst] += x * Y[sl_src]; acc["xz"][sl_dst] += x * Z[sl_src]
            acc["yy"][sl_dst] += y * Y[sl_src]; acc["yz"][sl_dst] += y * Z[sl_src]; acc["zz"][sl_dst] += z * Z[sl_src]
    n = acc["n"]
    nn = np.maximum(n, 1)
    mx, my, mz = acc["x"] / nn, acc["y"] / nn, acc["z"] / nn
    C = np.empty((H, W, 3, 3))
    C[..., 0, 0] = acc["xx"] / nn - mx * mx; C[..., 0, 1] = acc["xy"] / nn - mx * my; C[..., 0, 2] = acc["xz"] / nn - mx * mz
    C[..., 1, 1] = acc["yy"] / nn - my * my; C[..., 1, 2] = acc["yz"] / nn - my * mz; C[..., 2, 2] = acc["zz"] / nn - mz * mz
    C[..., 1, 0] = C[..., 0, 1]; C[..., 2, 0] = C[..., 0, 2]; C[..., 2, 1] = C[..., 1, 2]
    w, V = np.linalg.eigh(C)
    nrm = V[..., :, 0].copy()
    flip = (nrm * P).sum(-1) > 0                        # toward the camera: n . p < 0
    nrm[flip] *= -1
    bad = (n < 6) | ~valid | (w[..., 0] > 1e-5)
    nrm[bad] = np.nan
    return nrm.astype(np.float32), w[..., 0]


def depth_normals_gradient(depth_u16, K):
    """cv::rgbd::RgbdNormals(..., RGBD_NORMALS_METHOD_LINEMOD) on the raw 16-bit depth image (rgbd.cpp:199-205), restated from the
    published method (Hinterstoisser et al., PAMI 2012, section 2.4): least-squares depth gradient over the 8 neighbours at +-5
    pixels whose depth differs from the centre by less than 50 raw units; normal of the tangent plane through the back-projected
    X, X(x+1), X(y+1); integer sums, float cross product, normalised, pointed at the camera.  OpenCV absent: UNPINNED."""
    fx, cx, fy, cy = (np.float32(v) for v in K)
    D = np.asarray(depth_u16).astype(np.int64)
    H, W = D.shape
    r = 5
    A0 = np.zeros((H, W), np.int64); A1 = np.zeros_like(A0); A3 = np.zeros_like(A0); b0 = np.zeros_like(A0); b1 = np.zeros_like(A0)
    ys, xs = slice(r, H - r - 1), slice(r, W - r - 1)
    d = D[ys, xs]
    for j in (-r, 0, r):
        for i in (-r, 0, r):
            delta = D[r + j:H - r - 1 + j, r + i:W - r - 1 + i] - d
            ok = np.abs(delta) < 50
            A0[ys, xs] += ok * (i * i); A1[ys, xs] += ok * (i * j); A3[ys, xs] += ok * (j * j)
            b0[ys, xs] += np.where(ok, i * delta, 0); b1[ys, xs] += np.where(ok, j * delta, 0)
    det = A0 * A3 - A1 * A1
    gx = A3 * b0 - A1 * b1
    gy = -A1 * b0 + A0 * b1
    yy, xx = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    f = np.float32
    k00 = f(1.0) / fx; k02 = (f(0.0) * cy - cx * fy) / (fx * fy); k11 = f(1.0) / fy; k12 = -cy / fy
    a1 = (D * det + (xx + 1) * gx).astype(np.float32); b1f = (yy * gx).astype(np.float32); c1 = gx.astype(np.float32)
    a2 = (xx * gy).astype(np.float32); b2f = (D * det + (yy + 1) * gy).astype(np.float32); c2 = gy.astype(np.float32)
    X1x = k00 * a1 + (f(0.0) * b1f + k02 * c1); X1y = k11 * b1f + k12 * c1; X1z = c1
    X2x = k00 * a2 + (f(0.0) * b2f + k02 * c2); X2y = k11 * b2f + k12 * c2; X2z = c2
    nx = X1y * X2z - X1z * X2y; ny = X1z * X2x - X1x * X2z; nz = X1x * X2y - X1y * X2x
    ln = np.sqrt(nx.astype(np.float64) ** 2 + ny.astype(np.float64) ** 2 + nz.astype(np.float64) ** 2)
    out = np.full((H, W, 3), np.nan, np.float32)
    inner = np.zeros((H, W), bool); inner[ys, xs] = True
    good = inner & (ln > 0)
    s = np.where(nz > 0, -1.0, 1.0) / np.where(good, ln, 1.0)
    for k, c in enumerate((nx, ny, nz)):
        out[..., k] = np.where(good, (c.astype(np.float64) * s).astype(np.float32), np.float32(np.nan))
    return out


def ingest_scene(depth_u16, prob_u16, K, depth_scale, voxel=0.005, class_threshold=0.10, normal_method=0):
    """normal_method 0: depth-gradient normals (depth_normals_gradient), 1: the 5x5 plane fit (depth_normals; the golden fixtures)."""
    fx, cx, fy, cy = (np.float32(v) for v in K)
    d = depth_u16.astype(np.float32) * np.float32(depth_scale)
    H, W = d.shape
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    P = np.stack([((jj - np.float64(cx)) * d / np.float64(fx)), ((ii - np.float64(cy)) * d / np.float64(fy)), d], axis=-1).astype(np.float32)  # rgbd.cpp:214-216
    normals = depth_normals_gradient(depth_u16, K) if normal_method == 0 else depth_normals(P, d > 0)[0]
    cloud = voxel_grid(P.reshape(-1, 3), voxel)                                                  # :228-231
    radius = 2.0 * float(np.float32(voxel)) + 0.005                                              # :235
    tree = cKDTree(cloud.astype(np.float64))
    k = tree.query_ball_point(cloud.astype(np.float64), radius, return_length=True)
    keep = k > 10                                                                                # :236 (itself included)
    pos, nrm, pr, pix = [], [], [], []
    thr = np.float32(class_threshold)
    for pt in cloud[keep]:
        if not np.isfinite(pt[2]) or pt[2] <= 0 or pt[2] > 2.0:                                  # :243-244
            continue
        col = int((fx * pt[0] + cx * pt[2]) / pt[2])                                             # :251-253 (float32)
        row = int((fy * pt[1] + cy * pt[2]) / pt[2])
        if not (0 <= row < H and 0 <= col < W):
            continue
        cp = np.float32(float(prob_u16[row, col]) * (1.0 / 10000))                               # :255
        if cp < thr:
            continue
        n = normals[row, col]
        if not np.isfinite(n).all() or (n == 0).all():                                           # :264-267
            continue
        pos.append(pt); nrm.append(n); pr.append(cp); pix.append((row, col))
    return (np.array(pos, np.float32).reshape(-1, 3), np.array(nrm, np.float32).reshape(-1, 3), np.array(pr, np.float32),
            np.array(pix, np.int32).reshape(-1, 2))


def preprocess_model(raw_xyz, normal_radius, voxel, model_scale):
    pts = np.asarray(raw_xyz, np.float32).astype(np.float64)        # PLY vertices are float32 in PCL
    tree = cKDTree(pts)
    nb = tree.query_ball_point(pts, float(np.float32(normal_radius)))
    nrm = np.full(pts.shape, np.nan)
    for i, idx in enumerate(nb):                                    # pcl::NormalEstimation, radius search (rgbd.cpp:72-83)
        if len(idx) < 3:
            continue
        q = pts[idx] - pts[idx].mean(axis=0)
        w, V = np.linalg.eigh(q.T @ q)
        n = V[:, 0]
        if np.dot(n, -pts[i]) < 0:                                  # flipNormalTowardsViewpoint(0,0,0)
            n = -n
        nrm[i] = -n                                                 # stocs.cpp:47-52 -> away from the origin
    ok = np.isfinite(nrm).all(axis=1)
    cen, navg = voxel_grid(pts[ok].astype(np.float32), voxel, nrm[ok].astype(np.float32))        # stocs.cpp:54-57
    ln = np.linalg.norm(navg.astype(np.float64), axis=1)
    fin = np.isfinite(navg).all(axis=1) & (ln > 0)
    cen, navg, ln = cen[fin], navg[fin], ln[fin]
    navg = (navg.astype(np.float64) / ln[:, None]).astype(np.float32)                            # set_normal, point3d.hpp:43-45
    return (cen * np.float32(model_scale)).astype(np.float32), navg


def icp(src, tgt, tgt_nrm, max_iterations=5, max_corr=0.035):
    """Linearised point-to-plane ICP (the algorithm behind pcl::IterativeClosestPointWithNormals as used by
    clustering::point_to_plane_icp, reference src/pose_clustering.cpp:123-140).  Returns (T 4x4, n_corr)."""
    src = np.asarray(src, np.float32).astype(np.float64)
    tgt = np.asarray(tgt, np.float32).astype(np.float64)
    nrm = np.asarray(tgt_nrm, np.float32).astype(np.float64)
    tree = cKDTree(tgt)
    T = np.eye(4)
    ncorr = 0
    for _ in range(max_iterations):
        s = src @ T[:3, :3].T + T[:3, 3]
        d, j = tree.query(s)
        ok = d * d <= float(np.float32(max_corr)) ** 2
        ncorr = int(ok.sum())
        if ncorr < 6:
            break
        s, t, n = s[ok], tgt[j[ok]], nrm[j[ok]]
        A = np.concatenate([np.cross(s, n), n], axis=1)
        b = ((t - s) * n).sum(1)
        x = np.linalg.solve(A.T @ A, A.T @ b)
        ca, sa, cb, sb, cg, sg = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
        U = np.eye(4)
        U[:3, :3] = [[cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa], [sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa],
                     [-sb, cb * sa, cb * ca]]
        U[:3, 3] = x[3:]
        T = U @ T
    return T, ncorr
