"""Seeded synthetic inputs for the StoCS hot path.

No reference code is involved.  The metric configuration "Cm" is a 5 000-point model on a closed
analytic surface (oblate ellipsoid + a spherical bump, analytic outward normals) and a 20 000-point
scene = the model under a ground-truth pose (camera-facing part, noisy) + a tilted table plane and
three distractor boxes sampled on a 5 mm lattice.  "C5" scales the same generators to 50k / 200k.

All arrays are float32, C-contiguous.  Scene points carry (row, col) pixels from a pin-hole
projection with the YCB intrinsics of the reference driver (stocs_match_one_object.cpp:21).
"""
from __future__ import annotations

import dataclasses

import numpy as np

SEED_MODEL = 0x5EED0001
SEED_POSE = 0x5EED0002
SEED_CAND = 0x5EED0003
SEED_SCENE = 0x5EED0004

YCB_INTRINSICS = (1066.778, 312.986, 1067.487, 241.310)  # fx, cx, fy, cy


@dataclasses.dataclass
class Cloud:
    pos: np.ndarray  # (n,3) f32
    nrm: np.ndarray  # (n,3) f32 unit


@dataclasses.dataclass
class Scene:
    pos: np.ndarray   # (n,3) f32, camera frame, metres
    nrm: np.ndarray   # (n,3) f32 unit
    prob: np.ndarray  # (n,)  f32 class probability
    pixel: np.ndarray  # (n,2) i32 (row, col)
    n_object: int      # the first n_object points come from the object
    T_gt: np.ndarray   # (4,4) f64 model(raw frame) -> camera


def _fibonacci_sphere(n: int) -> np.ndarray:
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def make_model(n: int = 5000, seed: int = SEED_MODEL, scale: float = 1.0) -> Cloud:
    """Oblate ellipsoid (0.08, 0.08, 0.03) m + spherical bump r=0.02 m at (0.05, 0, 0.02)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    abc = np.array([0.08, 0.08, 0.03]) * scale
    bc = np.array([0.05, 0.0, 0.02]) * scale
    br = 0.02 * scale
    n0 = max(20 * n, 20000)
    u = _fibonacci_sphere(n0)
    # ellipsoid, area-uniform by rejection on the area element
    a, b, c = abc
    g = np.sqrt((b * c * u[:, 0]) ** 2 + (a * c * u[:, 1]) ** 2 + (a * b * u[:, 2]) ** 2)
    keep = rng.random(n0) < g / g.max()
    pe = u[keep] * abc
    ne = pe / (abc ** 2)
    ne /= np.linalg.norm(ne, axis=1, keepdims=True)
    inside_sphere = np.linalg.norm(pe - bc, axis=1) < br
    pe, ne = pe[~inside_sphere], ne[~inside_sphere]
    # bump sphere at the same areal density
    area_e = 4 * np.pi * ((((a * b) ** 1.6075 + (a * c) ** 1.6075 + (b * c) ** 1.6075) / 3) ** (1 / 1.6075))
    dens = keep.sum() / area_e
    nb0 = int(dens * 4 * np.pi * br * br) + 1
    us = _fibonacci_sphere(nb0)
    ps = us * br + bc
    inside_ell = ((ps / abc) ** 2).sum(axis=1) < 1.0
    ps, ns = ps[~inside_ell], us[~inside_ell]
    pos = np.concatenate([pe, ps])
    nrm = np.concatenate([ne, ns])
    perm = rng.permutation(len(pos))[:n]
    if len(perm) < n:
        raise ValueError("oversampling factor too small")
    pos, nrm = pos[perm], nrm[perm]
    pos = pos + rng.normal(0.0, 0.0003 * scale, pos.shape)  # jitter 0.3 mm
    pos = pos - pos.mean(axis=0)
    return Cloud(np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(nrm, np.float32))


def make_model_asym(n: int = 5000, seed: int = SEED_MODEL + 101) -> Cloud:
    """A model without any symmetry, for pose-quality reports: tri-axial ellipsoid (0.08, 0.055, 0.03) m with two spherical bumps of
    different size off every symmetry plane -- r = 0.02 m at (0.05, 0.01, 0.02) and r = 0.012 m at (-0.035, 0.03, -0.012).  Every
    rotation of it is observable (the metric model "Cm", SURVEY 8d, is an ellipsoid of revolution with one bump: a rotation about
    its axis barely changes it, so the rotation error of its winners says little).  Same sampling as make_model."""
    rng = np.random.Generator(np.random.PCG64(seed))
    abc = np.array([0.08, 0.055, 0.03])
    bumps = [(np.array([0.05, 0.01, 0.02]), 0.02), (np.array([-0.035, 0.03, -0.012]), 0.012)]
    n0 = max(20 * n, 20000)
    u = _fibonacci_sphere(n0)
    a, b, c = abc
    g = np.sqrt((b * c * u[:, 0]) ** 2 + (a * c * u[:, 1]) ** 2 + (a * b * u[:, 2]) ** 2)
    keep = rng.random(n0) < g / g.max()
    pe = u[keep] * abc
    ne = pe / (abc ** 2)
    ne /= np.linalg.norm(ne, axis=1, keepdims=True)
    area_e = 4 * np.pi * ((((a * b) ** 1.6075 + (a * c) ** 1.6075 + (b * c) ** 1.6075) / 3) ** (1 / 1.6075))
    dens = keep.sum() / area_e
    inside_any = np.zeros(len(pe), bool)
    for bc, br in bumps:
        inside_any |= np.linalg.norm(pe - bc, axis=1) < br
    parts_p, parts_n = [pe[~inside_any]], [ne[~inside_any]]
    for k, (bc, br) in enumerate(bumps):
        us = _fibonacci_sphere(int(dens * 4 * np.pi * br * br) + 1)
        ps = us * br + bc
        hidden = ((ps / abc) ** 2).sum(axis=1) < 1.0
        for j, (oc, orr) in enumerate(bumps):
            if j != k:
                hidden |= np.linalg.norm(ps - oc, axis=1) < orr
        parts_p.append(ps[~hidden]); parts_n.append(us[~hidden])
    pos = np.concatenate(parts_p); nrm = np.concatenate(parts_n)
    perm = rng.permutation(len(pos))[:n]
    if len(perm) < n:
        raise ValueError("oversampling factor too small")
    pos, nrm = pos[perm], nrm[perm]
    pos = pos + rng.normal(0.0, 0.0003, pos.shape)
    pos = pos - pos.mean(axis=0)
    return Cloud(np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(nrm, np.float32))


def random_rotation(rng: np.random.Generator) -> np.ndarray:
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def gt_pose(seed: int = SEED_POSE) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    T = np.eye(4)
    T[:3, :3] = random_rotation(rng)
    T[:3, 3] = [0.05, -0.03, 0.80]
    return T


def _perturb_normals(rng, nrm, max_deg):
    ang = np.deg2rad(max_deg) * rng.random(len(nrm))
    axis = rng.normal(size=nrm.shape)
    axis -= (axis * nrm).sum(1, keepdims=True) * nrm
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    out = nrm * np.cos(ang)[:, None] + axis * np.sin(ang)[:, None]
    return out / np.linalg.norm(out, axis=1, keepdims=True)


def _project(pos, intr=YCB_INTRINSICS, width=640, height=480):
    fx, cx, fy, cy = intr
    col = (pos[:, 0] * fx / pos[:, 2] + cx).astype(np.int64)
    row = (pos[:, 1] * fy / pos[:, 2] + cy).astype(np.int64)
    col = np.clip(col, 0, width - 1)
    row = np.clip(row, 0, height - 1)
    return np.stack([row, col], axis=1).astype(np.int32)


def make_scene(model: Cloud, n: int = 20000, seed: int = SEED_SCENE, T_gt: np.ndarray | None = None,
               lattice: float = 0.005, noise: float = 0.001) -> Scene:
    rng = np.random.Generator(np.random.PCG64(seed))
    if T_gt is None:
        T_gt = gt_pose()
    R, t = T_gt[:3, :3], T_gt[:3, 3]
    p = model.pos.astype(np.float64) @ R.T + t
    nn = model.nrm.astype(np.float64) @ R.T
    facing = (nn * (-p / np.linalg.norm(p, axis=1, keepdims=True))).sum(1) > 0
    p, nn = p[facing], nn[facing]
    # thin the object to at most n/8 points so clutter dominates ("noisy segmentation")
    max_obj = max(n // 8, 16)
    if len(p) > max_obj:
        sel = rng.permutation(len(p))[:max_obj]
        p, nn = p[sel], nn[sel]
    p = p + rng.normal(0.0, noise, p.shape)
    nn = _perturb_normals(rng, nn, 5.0)
    prob_obj = np.clip(rng.normal(0.85, 0.1, len(p)), 0.1, 1.0)
    n_obj = len(p)

    # clutter: tilted table + three boxes on a lattice
    th = np.deg2rad(20.0)
    e1 = np.array([1.0, 0.0, 0.0])
    e2 = np.array([0.0, np.sin(th), np.cos(th)])
    origin = t + np.array([0.0, 0.06, 0.0])
    k = int(np.ceil(np.sqrt(n) * 0.75))
    ii, jj = np.meshgrid(np.arange(-k, k + 1), np.arange(-k, k + 1), indexing="ij")
    order = np.argsort((ii.ravel() ** 2 + jj.ravel() ** 2), kind="stable")
    ij = np.stack([ii.ravel()[order], jj.ravel()[order]], axis=1).astype(np.float64)
    table = origin + lattice * (ij[:, :1] * e1 + ij[:, 1:] * e2)
    ntab = np.cross(e1, e2)
    boxes = []
    for bi, (cx, cz, sx, sy, sz) in enumerate([(-0.20, 0.05, 0.10, 0.08, 0.10), (0.22, -0.05, 0.08, 0.12, 0.08),
                                               (0.02, 0.22, 0.16, 0.06, 0.06)]):
        c0 = origin + cx * e1 + cz * e2 - ntab * 0.0
        nx, ny, nz = int(sx / lattice), int(sy / lattice), int(sz / lattice)
        up = -ntab if ntab[1] > 0 else ntab  # "up" is -y in camera frame
        # top face and camera-facing front face
        a_, b_ = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
        top = c0 + lattice * (a_.ravel()[:, None] * e1 + b_.ravel()[:, None] * e2) + up * sy
        a2, h2 = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
        front = c0 + lattice * (a2.ravel()[:, None] * e1) + up * (lattice * h2.ravel()[:, None])
        boxes.append((top, np.tile(up, (len(top), 1))))
        boxes.append((front, np.tile(-e2, (len(front), 1))))
    bpos = np.concatenate([b[0] for b in boxes])
    bnrm = np.concatenate([b[1] for b in boxes])
    need = n - n_obj - len(bpos)
    if need < 0:
        bpos, bnrm = bpos[: n - n_obj], bnrm[: n - n_obj]
        need = 0
    if need > len(table):
        raise ValueError("table lattice too small")
    cpos = np.concatenate([bpos, table[:need]])
    cnrm = np.concatenate([bnrm, np.tile(ntab, (need, 1))])
    flip = (cnrm * (-cpos)).sum(1) < 0
    cnrm[flip] *= -1.0
    cpos = cpos + rng.normal(0.0, noise * 0.2, cpos.shape)
    prob_c = rng.uniform(0.10, 0.45, len(cpos))

    pos = np.concatenate([p, cpos]).astype(np.float32)
    nrm = np.concatenate([nn, cnrm]).astype(np.float32)
    prob = np.concatenate([prob_obj, prob_c]).astype(np.float32)
    return Scene(np.ascontiguousarray(pos), np.ascontiguousarray(nrm), np.ascontiguousarray(prob),
                 _project(pos.astype(np.float64)), n_obj, T_gt)


def _rot_axis_angle(axis, ang):
    axis = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
    x, y, z = axis[..., 0], axis[..., 1], axis[..., 2]
    c, s = np.cos(ang), np.sin(ang)
    C = 1 - c
    R = np.empty(axis.shape[:-1] + (3, 3))
    R[..., 0, 0] = c + x * x * C
    R[..., 0, 1] = x * y * C - z * s
    R[..., 0, 2] = x * z * C + y * s
    R[..., 1, 0] = y * x * C + z * s
    R[..., 1, 1] = c + y * y * C
    R[..., 1, 2] = y * z * C - x * s
    R[..., 2, 0] = z * x * C - y * s
    R[..., 2, 1] = z * y * C + x * s
    R[..., 2, 2] = c + z * z * C
    return R


def make_candidates(T_centred_gt: np.ndarray, k: int = 65536, seed: int = SEED_CAND) -> np.ndarray:
    """K candidate transforms in the CENTRED frames (what the LCP kernel scores), column-major
    (k,16) f32: 1 % within (1 mm, 1 deg) of the ground truth, 9 % within (1 cm, 5 deg), 90 % within
    (5 cm, 30 deg), perturbed about the object centroid (the origin of the centred model)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tier = rng.random(k)
    max_t = np.where(tier < 0.01, 0.001, np.where(tier < 0.10, 0.01, 0.05))
    max_r = np.deg2rad(np.where(tier < 0.01, 1.0, np.where(tier < 0.10, 5.0, 30.0)))
    axis = rng.normal(size=(k, 3))
    ang = max_r * rng.random(k)
    dR = _rot_axis_angle(axis, ang)
    d = rng.normal(size=(k, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dt = d * (max_t * rng.random(k) ** (1 / 3))[:, None]
    R0, t0 = T_centred_gt[:3, :3], T_centred_gt[:3, 3]
    R = np.einsum("ij,kjl->kil", R0, dR)  # perturb about the model origin
    T = np.zeros((k, 4, 4))
    T[:, :3, :3] = R
    T[:, :3, 3] = t0 + dt
    T[:, 3, 3] = 1.0
    return np.ascontiguousarray(T.transpose(0, 2, 1).reshape(k, 16).astype(np.float32))  # column-major


def centred_gt(T_gt: np.ndarray, centroid_scene: np.ndarray, centroid_model: np.ndarray) -> np.ndarray:
    """Ground-truth transform between the centroid-shifted clouds (stocs.cpp:943-964)."""
    T = np.eye(4)
    T[:3, :3] = T_gt[:3, :3]
    T[:3, 3] = T_gt[:3, :3] @ centroid_model + T_gt[:3, 3] - centroid_scene
    return T


def workload(name: str = "Cm"):
    """Returns (model Cloud, Scene, n_candidates) for the named configuration."""
    if name == "Cm":
        m = make_model(5000)
        return m, make_scene(m, 20000), 65536
    if name == "Cm_asym":   # the metric sizes with a model without symmetry (pose-quality reports; not the metric workload)
        m = make_model_asym(5000)
        return m, make_scene(m, 20000, seed=SEED_SCENE + 101), 65536
    if name == "C5":
        m = make_model(50000)
        return m, make_scene(m, 200000, lattice=0.0016), 16384
    if name == "tiny":
        m = make_model(400, seed=SEED_MODEL + 7)
        return m, make_scene(m, 1500, seed=SEED_SCENE + 7), 256
    if name == "dense":
        # C5-like density at a CPU-checkable size: ~1.6 mm scene spacing -> ~60 candidates per grid list,
        # which switches the library to the epsilon/2 grid and the deep-unroll scan
        m = make_model(3000, seed=SEED_MODEL + 13, scale=0.5)
        return m, make_scene(m, 24000, seed=SEED_SCENE + 13, lattice=0.0016), 512
    if name == "small":
        m = make_model(1000, seed=SEED_MODEL + 11)
        return m, make_scene(m, 5000, seed=SEED_SCENE + 11), 2048
    raise KeyError(name)
