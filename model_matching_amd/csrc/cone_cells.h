// cone_cells.h -- the direction cells "coloured" by one cone query of the normal set.
// Restates Super4PCS::IndexedNormalSet::getNeighbors(p, n, cosAlpha) (reference
// include/super4pcs/accelerators/normalset.hpp:166-214) up to the point where the set of direction cells is known:
//   q = Quaternion::setFromTwoVectors(z, n);  for every sample a:  dir = (q * d_a).normalized();
//   id = indexNormal(dir) = int((dir/2 + 0.5) / _nepsilon) per axis, base 7 (normalset.h:100-104)
// Shared by host and device (and exercised on the host by tests/test_abi_cpu.py through stocs_cone_cells_host).
//
// Two evaluations of the same cell:
//   * exact  -- the reference's float operations in the reference's order (IEEE divide / sqrt, no contraction):
//               ~110 instructions per sample on gfx950;
//   * filter -- the same map in matrix form with FMAs and a reciprocal square root (~35 instructions per sample) that also returns how far the sample is
//               from the nearest cell boundary.  Both evaluations approximate the same real number with an absolute error
//               below 1e-5 cell units (float32, |t| < 7, a dozen operations each), so when the filtered value is more
//               than CONE_MARGIN = 5e-5 away from every integer the two truncations agree and the cheap one is used; in
//               the remaining ~3e-4 of the samples (and for anything non-finite) the exact evaluation decides.
// The result is therefore bit-identical to the exact evaluation alone (asserted over 10^7 random samples on the host,
// and implicitly by every GPU-vs-oracle quad comparison).
#ifndef STOCS_CONE_CELLS_H
#define STOCS_CONE_CELLS_H

#include "stocs_math.h"

namespace stocs {

#define STOCS_MAX_CONE 64
#define STOCS_CONE_MARGIN 5e-5f

// normalset.h:100-104 + utils.h:139-148: int truncation, x fastest
STOCS_HD int index_normal(V3 n, float nepsilon) {
    const V3 half = mk3(0.5f, 0.5f, 0.5f);
    const V3 cn = (n / 2.0f + half) / nepsilon;
    return (int)cn.z * 49 + ((int)cn.y * 7 + (int)cn.x);
}

// Eigen Quaternion::setFromTwoVectors((0,0,1), n) and q * v (DESIGN.md "numerics")
STOCS_HD void quat_from_z(V3 n, float q[4]) {
    const V3 v0 = normalized3(mk3(0.f, 0.f, 1.f));
    const V3 v1 = normalized3(n);
    float c = dot3(v1, v0);
    if (c < -1.0f + 1e-5f) {
        // Eigen takes the axis from an SVD here (Quaternion.h: the right-singular vector of the 2x3 matrix [v0; v1] that belongs to
        // the vanishing singular value, i.e. the unit vector orthogonal to both).  Restated as what that vector IS -- normalize(v0 x v1),
        // signed so that the rotation by acos(c) about it takes v0 to v1; (1,0,0) when the two are exactly opposite and every vector
        // orthogonal to v0 qualifies.  The sign the library's Jacobi sweeps would leave on it, and its choice in the exactly opposite
        // case, cannot be pinned without the library (parity unpinned; the case has probability ~1e-5 per query normal).
        c = c > -1.0f ? c : -1.0f;
        const float w2 = (1.0f + c) * 0.5f;
        const float s = stocs_sqrtf(1.0f - w2);
        const V3 x = cross3(v0, v1);
        const float x2 = dot3(x, x);
        const V3 ax = x2 > 0.0f ? x / stocs_sqrtf(x2) : mk3(1.0f, 0.0f, 0.0f);
        q[0] = ax.x * s; q[1] = ax.y * s; q[2] = ax.z * s; q[3] = stocs_sqrtf(w2);
        return;
    }
    const V3 axis = cross3(v0, v1);
    const float s = stocs_sqrtf((1.0f + c) * 2.0f);
    const float invs = 1.0f / s;
    q[0] = axis.x * invs; q[1] = axis.y * invs; q[2] = axis.z * invs; q[3] = s * 0.5f;
}
STOCS_HD V3 quat_rot(const float q[4], V3 v) {
    const V3 qv = mk3(q[0], q[1], q[2]);
    V3 uv = cross3(qv, v);
    uv = uv + uv;
    return (v + q[3] * uv) + cross3(qv, uv);
}

// exact cell of sample d under q: normalset.hpp:186-197; -1 when std::array::at would throw (NaN direction)
STOCS_HD int cone_cell_exact(const float q[4], V3 d, float nepsilon) {
    const V3 dir = normalized3(quat_rot(q, d));
    const int id = index_normal(dir, nepsilon);
    return (id < 0 || id >= 343) ? -1 : id;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define STOCS_FMAF(a, b, c) __fmaf_rn((a), (b), (c))
#else
#define STOCS_FMAF(a, b, c) fmaf((a), (b), (c))
#endif

// Per-query constants of the filter.  q * v expands to M(q) v with M(q) = (1 - 2|qv|^2) I + 2 w [qv]x + 2 qv qv^T (the usual
// rotation-matrix entries, valid as an identity for ANY q -- near the anti-parallel case q is off unit length by up to
// 1e-3, and the reference's normalisation of the rotated sample then matters, so the filter normalises too).  The sample's
// z component (cos alpha, the same for every sample of a base) is folded into the constant column.
struct ConeFilter {
    float ax, ay, az, bx, by, bz, cx, cy, cz, s;
};
STOCS_HD ConeFilter cone_filter_setup(const float q[4], float dz, float half_inv_neps) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    ConeFilter f;
    f.ax = 1.0f - 2.0f * (y * y + z * z); f.ay = 2.0f * (x * y + w * z); f.az = 2.0f * (x * z - w * y);
    f.bx = 2.0f * (x * y - w * z); f.by = 1.0f - 2.0f * (x * x + z * z); f.bz = 2.0f * (y * z + w * x);
    f.cx = dz * (2.0f * (x * z + w * y)); f.cy = dz * (2.0f * (y * z - w * x)); f.cz = dz * (1.0f - 2.0f * (x * x + y * y));
    f.s = half_inv_neps;
    return f;
}
#if defined(__HIP_DEVICE_COMPILE__)
#define STOCS_RSQRTF(a) __frsqrt_rn(a)
#else
#define STOCS_RSQRTF(a) (1.0f / sqrtf(a))
#endif
// cell of the sample with (x, y) components (dx, dy), or -1 when the filter cannot decide (use cone_cell_exact then)
STOCS_HD int cone_cell_filtered(const ConeFilter& f, float dx, float dy) {
    const float vx = STOCS_FMAF(dx, f.ax, STOCS_FMAF(dy, f.bx, f.cx));
    const float vy = STOCS_FMAF(dx, f.ay, STOCS_FMAF(dy, f.by, f.cy));
    const float vz = STOCS_FMAF(dx, f.az, STOCS_FMAF(dy, f.bz, f.cz));
    const float k = STOCS_RSQRTF(STOCS_FMAF(vx, vx, STOCS_FMAF(vy, vy, vz * vz))) * f.s;
    const float tx = STOCS_FMAF(vx, k, f.s), ty = STOCS_FMAF(vy, k, f.s), tz = STOCS_FMAF(vz, k, f.s);
    const float fx = floorf(tx), fy = floorf(ty), fz = floorf(tz);
    const float rx = tx - fx, ry = ty - fy, rz = tz - fz;
    const float lo = fminf(rx, fminf(ry, rz)), hi = fmaxf(rx, fmaxf(ry, rz));
    // t = s (v/|v| + 1) lies in [-1e-6, 2 s + 1e-6] = [-1e-6, 6.99952]: a coordinate below 0 has a fractional part just under
    // 1 and fails the `hi` test, so a certain sample has every cell coordinate in 0..6.  Every comparison is false for a
    // NaN, so non-finite samples take the exact path.
    const bool certain = lo > STOCS_CONE_MARGIN && hi < 1.0f - STOCS_CONE_MARGIN;
    return certain ? (int)fz * 49 + ((int)fy * 7 + (int)fx) : -1;
}

}  // namespace stocs
#endif
