// cluster.hip -- host-side pose clustering.  Replaces clustering::greedy_clustering with
// get_pose_diff and quaternion_to_euler (reference src/pose_clustering.cpp:79-121, 27-71, 5-25).
// O(K log K + K*C) on at most 20 000 scored candidates: stays on the host, as in the reference
// (where it has no caller at all).  std::sort in the reference is unstable; a stable sort by
// descending score is used so that results are reproducible.
#include <math.h>

#include <algorithm>
#include <vector>

#include "stocs_ctx.h"

namespace stocs {

struct HM3 { float m[3][3]; };
static HM3 hmul(const HM3& A, const HM3& B) {
    HM3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + (A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j]);
    return C;
}
static HM3 hinverse(const HM3& a) {  // cofactor inverse, as Eigen does for fixed 3x3
    HM3 r;
    const float c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
    const float c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
    const float c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    const float det = a.m[0][0] * c00 + (a.m[0][1] * c01 + a.m[0][2] * c02);
    const float inv = 1.0f / det;
    r.m[0][0] = c00 * inv; r.m[1][0] = c01 * inv; r.m[2][0] = c02 * inv;
    r.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * inv;
    r.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * inv;
    r.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * inv;
    r.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * inv;
    r.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * inv;
    r.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * inv;
    return r;
}
static void mat_to_quat(const HM3& a, float q[4] /*x,y,z,w*/) {  // Eigen Quaternion(Matrix3)
    float t = a.m[0][0] + a.m[1][1] + a.m[2][2];
    if (t > 0.0f) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (a.m[2][1] - a.m[1][2]) * t;
        q[1] = (a.m[0][2] - a.m[2][0]) * t;
        q[2] = (a.m[1][0] - a.m[0][1]) * t;
    } else {
        int i = 0;
        if (a.m[1][1] > a.m[0][0]) i = 1;
        if (a.m[2][2] > a.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (a.m[k][j] - a.m[j][k]) * t;
        q[j] = (a.m[j][i] + a.m[i][j]) * t;
        q[k] = (a.m[k][i] + a.m[i][k]) * t;
    }
}
static void pose_diff(const float* test, const float* base, const float* sym, float& rot_err, float& tr_err) {
    HM3 t, b;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { t.m[i][j] = test[j * 4 + i]; b.m[i][j] = base[j * 4 + i]; }
    const HM3 diff = hmul(hinverse(t), b);
    float q[4], e[3];
    mat_to_quat(diff, q);
    // quaternion_to_euler, pose_clustering.cpp:5-25 (float products promoted to double)
    const double sinr = +2.0 * (double)(q[3] * q[0] + q[1] * q[2]);
    const double cosr = +1.0 - 2.0 * (double)(q[0] * q[0] + q[1] * q[1]);
    e[0] = (float)atan2(sinr, cosr);
    const double sinp = +2.0 * (double)(q[3] * q[1] - q[2] * q[0]);
    if (fabs(sinp) >= 1) e[1] = (float)copysign(M_PI / 2, sinp);
    else e[1] = (float)asin(sinp);
    const double siny = +2.0 * (double)(q[3] * q[2] + q[0] * q[1]);
    const double cosy = +1.0 - 2.0 * (double)(q[1] * q[1] + q[2] * q[2]);
    e[2] = (float)atan2(siny, cosy);
    for (int d = 0; d < 3; ++d) {
        e[d] = (float)((double)e[d] * 180.0 / M_PI);
        e[d] = fabsf(e[d]);
        if (sym[d] == 90) {
            e[d] = fabsf(e[d] - 90);
            e[d] = std::min(e[d], 90 - e[d]);
        } else if (sym[d] == 180) {
            e[d] = std::min(e[d], 180 - e[d]);
        } else if (sym[d] == 360) {
            e[d] = 0;
        }
    }
    rot_err = std::max(std::max(e[0], e[1]), e[2]);
    tr_err = (float)sqrt(pow((double)(base[12] - test[12]), 2) + pow((double)(base[13] - test[13]), 2) +
                         pow((double)(base[14] - test[14]), 2));
}

}  // namespace stocs

using namespace stocs;

extern "C" int stocs_cluster_poses(const float* poses16, const float* lcp, int n, float acceptable_fraction, float best_score,
                                   int maximum_pose_count, float min_distance, float min_angle, const float* sym3,
                                   int32_t* out_idx, int cap, int* n_out) {
    if (n < 0 || (n && (!poses16 || !lcp)) || !sym3 || !n_out) return STOCS_ERR_INVALID;
    std::vector<int> pruned;
    for (int i = 0; i < n; ++i)
        if (lcp[i] > acceptable_fraction * best_score) pruned.push_back(i);
    std::stable_sort(pruned.begin(), pruned.end(), [&](int a, int b) { return lcp[a] > lcp[b]; });
    std::vector<int> kept;
    for (size_t ci = 0; ci < pruned.size(); ++ci) {
        const int cand = pruned[ci];
        bool inValid = false;
        for (size_t k = 0; k < kept.size(); ++k) {
            float re, te;
            pose_diff(poses16 + (size_t)cand * 16, poses16 + (size_t)kept[k] * 16, sym3, re, te);
            if (re < min_angle && te < min_distance) { inValid = true; break; }
        }
        if (!inValid) kept.push_back(cand);
        if ((int)kept.size() > maximum_pose_count) break;  // sic: size > count
    }
    *n_out = (int)kept.size();
    for (size_t i = 0; i < kept.size() && (int)i < cap; ++i) out_idx[i] = kept[i];
    return ((int)kept.size() > cap && out_idx) ? STOCS_ERR_CAPACITY : STOCS_OK;
}
