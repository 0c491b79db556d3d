/*
 * stocs_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT.  See stocs_oracle.h.
 * PARITY STATUS: "parity unpinned" (no reference tests / golden vectors exist; reference not buildable).
 *
 * Conventions for arithmetic that lives in the (absent) Eigen dependency -- stated once, used
 * everywhere, also documented in DESIGN.md "numerics":
 *   - all vectors float; 3-term sums (dot, squaredNorm, 3x3*3, 3x3*3x3) are  e0 + (e1 + e2)
 *     (Eigen redux_novec_unroller halves the range: {0} + {1,2});
 *   - normalized(): z = squaredNorm; z > 0 ? v / sqrt(z) : v   (Eigen 3.3 MatrixBase::normalized);
 *   - 4x4 * homogeneous(vec3): ((m_i0*x + m_i1*y) + m_i2*z) + m_i3;
 *   - no FMA contraction anywhere (-ffp-contract=off), IEEE divide and sqrt.
 * Build: g++ -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "stocs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include <unordered_map>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ----------------------------------------------------------------------------------------------
// tiny vector algebra with the operation order stated in the header comment
// ----------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 mk(float x, float y, float z) { V3 v = {x, y, z}; return v; }
inline V3 ld(const float* p) { return mk(p[0], p[1], p[2]); }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline float sqn(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return sqrtf(sqn(a)); }
inline V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline V3 normalized(V3 a) {
    float z = sqn(a);
    if (z > 0.0f) return a / sqrtf(z);
    return a;
}

struct M3 { float m[3][3]; };  // m[row][col]
inline V3 mul(const M3& A, V3 v) {
    return mk(A.m[0][0] * v.x + (A.m[0][1] * v.y + A.m[0][2] * v.z),
              A.m[1][0] * v.x + (A.m[1][1] * v.y + A.m[1][2] * v.z),
              A.m[2][0] * v.x + (A.m[2][1] * v.y + A.m[2][2] * v.z));
}
inline M3 mul(const M3& A, const M3& B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[i][j] = A.m[i][0] * B.m[0][j] + (A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j]);
    return C;
}

// column-major 4x4 as in Eigen::Matrix4f::data(): T[c*4+r]
inline float T_at(const float* T, int r, int c) { return T[c * 4 + r]; }
inline V3 xform_point(const float* T, V3 p) {
    return mk(((T_at(T, 0, 0) * p.x + T_at(T, 0, 1) * p.y) + T_at(T, 0, 2) * p.z) + T_at(T, 0, 3),
              ((T_at(T, 1, 0) * p.x + T_at(T, 1, 1) * p.y) + T_at(T, 1, 2) * p.z) + T_at(T, 1, 3),
              ((T_at(T, 2, 0) * p.x + T_at(T, 2, 1) * p.y) + T_at(T, 2, 2) * p.z) + T_at(T, 2, 3));
}
inline V3 xform_normal(const float* T, V3 n) {
    return mk(T_at(T, 0, 0) * n.x + (T_at(T, 0, 1) * n.y + T_at(T, 0, 2) * n.z),
              T_at(T, 1, 0) * n.x + (T_at(T, 1, 1) * n.y + T_at(T, 1, 2) * n.z),
              T_at(T, 2, 0) * n.x + (T_at(T, 2, 1) * n.y + T_at(T, 2, 2) * n.z));
}

// ----------------------------------------------------------------------------------------------
// Point record: include/point3d.hpp:11-92 (pos, normal, pixel(row,col), class/current probability)
// ----------------------------------------------------------------------------------------------
struct Pt {
    V3 pos, nrm;
    int row, col;
    float class_prob;  // class_probability_
    float cur_prob;    // current_probability_
};

// ----------------------------------------------------------------------------------------------
// row 1: PPF -- src/rgbd.cpp:85-97 (ppf_closest_bin), 99-121 (ppf_compute)
// ----------------------------------------------------------------------------------------------
int closest_bin(int value, int discretization) {
    int lower_limit = value - (value % discretization);
    int upper_limit = lower_limit + discretization;
    int dist_from_lower = value - lower_limit;
    int dist_from_upper = upper_limit - value;
    return (dist_from_lower < dist_from_upper) ? lower_limit : upper_limit;
}

// int(x) of a double that may be NaN / out of range: mirror x86 cvttsd2si (INT_MIN), which is
// what the reference binary would produce; such keys are never in the map.
inline int trunc_int(double d) {
    if (!(d > -2147483648.0 && d < 2147483648.0)) return std::numeric_limits<int>::min();
    return (int)d;
}

void ppf_compute(V3 p1, V3 n1, V3 p2, V3 n2, int tr, int rot, int mode, int* out) {
    V3 u = p1 - p2;  // rgbd.cpp:110 (same u for both normals, not negated)
    int f0 = trunc_int((double)(norm(u) * 1000));  // float*int -> float, :112
    int f1, f2, f3;
    if (mode == 0) {
        // atan2(float,float) resolved to ::atan2(double,double): arguments promoted, double result
        f1 = trunc_int(atan2((double)norm(cross(n1, u)), (double)dot(n1, u)) * 180 / M_PI);
        f2 = trunc_int(atan2((double)norm(cross(n2, u)), (double)dot(n2, u)) * 180 / M_PI);
        f3 = trunc_int(atan2((double)norm(cross(n1, n2)), (double)dot(n1, n2)) * 180 / M_PI);
    } else {
        // float overload: atan2f, float*180 (int->float), then / M_PI in double
        f1 = trunc_int((double)(atan2f(norm(cross(n1, u)), dot(n1, u)) * 180) / M_PI);
        f2 = trunc_int((double)(atan2f(norm(cross(n2, u)), dot(n2, u)) * 180) / M_PI);
        f3 = trunc_int((double)(atan2f(norm(cross(n1, n2)), dot(n1, n2)) * 180) / M_PI);
    }
    out[0] = closest_bin(f0, tr);
    out[1] = closest_bin(f1, rot);
    out[2] = closest_bin(f2, rot);
    out[3] = closest_bin(f3, rot);
}

// ----------------------------------------------------------------------------------------------
// rows 2 / T4: the PPF index.
// Literal form (rgbd.cpp:123-154): every ordered pair is inserted under up to 128 keys.
// Query-side form used for big models: store each pair once under its own key F; lookup(K) is the
// union over the 128 offsets o of bucket(K - o), emitted in insertion order == lexicographic
// (id1,id2) (outer/inner loops stocs.cpp:63-64).  Equivalence is tested against the literal form.
// ----------------------------------------------------------------------------------------------
typedef std::pair<int, int> IPair;

struct KeyHash {
    size_t operator()(const std::array<int, 4>& k) const {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < 4; ++i) { h ^= (uint32_t)k[i]; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

}  // namespace

struct orc_index {
    int tr, rot;
    std::unordered_map<std::array<int, 4>, std::vector<IPair>, KeyHash> base;  // key F -> pairs
    // the key set of the reference's map (every F expanded by the 128 insertion offsets,
    // rgbd.cpp:130-137): what ppf_map.find(K) != end() tests
    std::unordered_map<std::array<int, 4>, char, KeyHash> keys;
    int64_t npairs;
};

struct orc_index_lit {
    std::map<std::vector<int>, std::vector<IPair> > map;  // PPFMapType, rgbd.hpp:23
};

namespace {

// `feat` (optional) receives, parallel to `out`, the quantised feature F each pair is stored under: (F, pair) is the
// pair's position in the product's index, which the walk-order enumeration of find_congruent sorts by.
int64_t index_lookup(const orc_index* ix, const int* K, std::vector<IPair>* out, std::vector<std::array<int, 4> >* feat = NULL) {
    if (out) out->clear();
    if (feat) feat->clear();
    std::vector<std::pair<IPair, std::array<int, 4> > > both;
    const int tr = ix->tr, rot = ix->rot;
    // rgbd.cpp:136: keys with p1 <= 5 or any angle key < 0 are never stored
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return 0;
    int64_t total = 0;
    // insertion offsets o0 in {-tr,0}, oi in {-2rot,-rot,0,rot}  =>  F = K - o
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b)
            for (int c = 0; c < 4; ++c)
                for (int d = 0; d < 4; ++d) {
                    std::array<int, 4> F = {{K[0] + a * tr, K[1] + (2 - b) * rot,
                                             K[2] + (2 - c) * rot, K[3] + (2 - d) * rot}};
                    auto it = ix->base.find(F);
                    if (it == ix->base.end()) continue;
                    total += (int64_t)it->second.size();
                    if (out && !feat) out->insert(out->end(), it->second.begin(), it->second.end());
                    if (out && feat) for (size_t k = 0; k < it->second.size(); ++k) both.push_back(std::make_pair(it->second[k], F));
                }
    if (out && !feat) std::sort(out->begin(), out->end());  // insertion order of the literal map
    if (out && feat) {   // same order (a pair occurs once: the pair alone decides), features alongside
        std::sort(both.begin(), both.end());
        for (size_t k = 0; k < both.size(); ++k) { out->push_back(both[k].first); feat->push_back(both[k].second); }
    }
    return total;
}

bool index_exists(const orc_index* ix, const int* K) {
    const int tr = ix->tr, rot = ix->rot;
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return false;
    if (!ix->keys.empty()) {
        std::array<int, 4> key = {{K[0], K[1], K[2], K[3]}};
        return ix->keys.find(key) != ix->keys.end();
    }
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b)
            for (int c = 0; c < 4; ++c)
                for (int d = 0; d < 4; ++d) {
                    std::array<int, 4> F = {{K[0] + a * tr, K[1] + (2 - b) * rot,
                                             K[2] + (2 - c) * rot, K[3] + (2 - d) * rot}};
                    if (ix->base.find(F) != ix->base.end()) return true;
                }
    return false;
}

// ----------------------------------------------------------------------------------------------
// rows 14/15: kd-tree -- include/super4pcs/accelerators/kdtree.h
//   finalize 355-370, createTree 560-641, split 522-538, doQueryRestrictedClosestIndex 394-459
//   AABB: bbox.h:65-107
// ----------------------------------------------------------------------------------------------
struct AABB {
    V3 mn, mx;
    AABB() {
        float h = std::numeric_limits<float>::max() / 2;  // bbox.h:65-66
        mn = mk(h, h, h);
        mx = mk(-h, -h, -h);
    }
    void extendTo(V3 q) {  // bbox.h:76-78
        if (q.x < mn.x) mn.x = q.x;
        if (q.y < mn.y) mn.y = q.y;
        if (q.z < mn.z) mn.z = q.z;
        if (q.x > mx.x) mx.x = q.x;
        if (q.y > mx.y) mx.y = q.y;
        if (q.z > mx.z) mx.z = q.z;
    }
    V3 center() const { return mn + ((mx - mn) / 2.0f); }  // bbox.h:91-92
};

struct KdNode {
    float splitValue;
    unsigned firstChildId;
    unsigned dim;
    unsigned leaf;
    unsigned start;
    unsigned size;
};

struct KdTree {
    std::vector<V3> pts;    // mPoints (physically partitioned)
    std::vector<int> idx;   // mIndices
    std::vector<KdNode> nodes;
    unsigned cell, maxDepth;

    KdTree() : cell(64), maxDepth(32) {}  // KD_POINT_PER_CELL 64, KD_MAX_DEPTH 32 (kdtree.h:60,63)

    void add(V3 p) { pts.push_back(p); idx.push_back((int)idx.size()); }  // :190-196

    unsigned split(int start, int end, unsigned dim, float splitValue) {  // :522-538
        int l(start), r(end - 1);
        for (; l < r; ++l, --r) {
            while (l < end && pts[l][dim] < splitValue) l++;
            while (r >= start && pts[r][dim] >= splitValue) r--;
            if (l > r) break;
            std::swap(pts[l], pts[r]);
            std::swap(idx[l], idx[r]);
        }
        return (pts[l][dim] < splitValue ? l + 1 : l);
    }

    void createTree(unsigned nodeId, unsigned start, unsigned end, unsigned level) {  // :560-641
        AABB aabb;
        for (unsigned i = start; i < end; ++i) aabb.extendTo(pts[i]);
        V3 diag = 0.5f * (aabb.mx - aabb.mn);
        unsigned dim = 0;  // maxCoeff: first maximum wins
        float best = diag.x;
        if (diag.y > best) { best = diag.y; dim = 1; }
        if (diag.z > best) { best = diag.z; dim = 2; }
        nodes[nodeId].dim = dim;
        nodes[nodeId].splitValue = aabb.center()[dim];
        unsigned midId = split(start, end, dim, nodes[nodeId].splitValue);
        nodes[nodeId].firstChildId = (unsigned)nodes.size();
        {
            KdNode n;
            memset(&n, 0, sizeof(n));
            nodes.push_back(n);
            nodes.push_back(n);
        }
        {
            unsigned childId = nodes[nodeId].firstChildId;
            if (midId - start <= cell || level >= maxDepth) {
                nodes[childId].leaf = 1;
                nodes[childId].start = start;
                nodes[childId].size = midId - start;
            } else {
                nodes[childId].leaf = 0;
                createTree(childId, start, midId, level + 1);
            }
        }
        {
            unsigned childId = nodes[nodeId].firstChildId + 1;
            if (end - midId <= cell || level >= maxDepth) {
                nodes[childId].leaf = 1;
                nodes[childId].start = midId;
                nodes[childId].size = end - midId;
            } else {
                nodes[childId].leaf = 0;
                createTree(childId, midId, end, level + 1);
            }
        }
    }

    void finalize() {  // :355-370
        nodes.clear();
        nodes.reserve(4 * pts.size() / cell + 16);
        KdNode root;
        memset(&root, 0, sizeof(root));
        nodes.push_back(root);
        nodes.back().leaf = 0;
        createTree(0, 0, (unsigned)pts.size(), 1);
    }

    struct QueryNode { unsigned nodeId; float sq; };

    // :394-459; the member stack mNodeStack[64] (:311) is made a local so OpenMP threads can share
    // the tree (Q15)
    int queryRestrictedClosestIndex(V3 q, float sqdist, int currentId = -1) const {
        QueryNode stack[64];
        int cl_id = -1;
        float cl_dist = sqdist;
        stack[0].nodeId = 0;
        stack[0].sq = 0.f;
        unsigned count = 1;
        while (count) {
            QueryNode& qnode = stack[count - 1];
            const KdNode& node = nodes[qnode.nodeId];
            if (qnode.sq < cl_dist) {
                if (node.leaf) {
                    --count;
                    const int end = (int)(node.start + node.size);
                    for (int i = (int)node.start; i < end; ++i) {
                        const float d = sqn(q - pts[i]);
                        if (d <= cl_dist && idx[i] != currentId) {  // inclusive, later wins ties (Q11)
                            cl_dist = d;
                            cl_id = idx[i];
                        }
                    }
                } else {
                    const float new_off = q[node.dim] - node.splitValue;
                    if (new_off < 0.) {
                        stack[count].nodeId = node.firstChildId;
                        qnode.nodeId = node.firstChildId + 1;
                    } else {
                        stack[count].nodeId = node.firstChildId + 1;
                        qnode.nodeId = node.firstChildId;
                    }
                    stack[count].sq = qnode.sq;
                    qnode.sq = new_off * new_off;
                    ++count;
                }
            } else {
                --count;
            }
        }
        return cl_id;
    }
};

// ----------------------------------------------------------------------------------------------
// the seeded draw that replaces stocs.cpp:133-148 (divergence Q6)
// ----------------------------------------------------------------------------------------------
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t rng64(uint64_t seed, uint64_t attempt, uint64_t k) {
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull);
    z = mix64(z ^ (attempt * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull));
    z = mix64(z ^ ((k + 1) * 0xDB4F0B9175AE2165ull));
    return z;
}
inline uint64_t wfix(float w) {
    if (!(w > 0.0f)) return 0;
    double s = (double)w * 4294967296.0;
    if (s >= 1.8446744073709552e19) return 0xFFFFFFFFFFFFFFFFull;
    return (uint64_t)s;
}
int draw_index(const float* w, int n, uint64_t r64) {
    uint64_t total = 0;
    for (int i = 0; i < n; ++i) total += wfix(w[i]);
    if (total == 0) return -1;
    uint64_t r = (uint64_t)(((unsigned __int128)r64 * (unsigned __int128)total) >> 64);
    uint64_t c = 0;
    for (int i = 0; i < n; ++i) {
        c += wfix(w[i]);
        if (c > r) return i;
    }
    return -1;
}

// ----------------------------------------------------------------------------------------------
// row 6: segment_distance_and_invariants -- stocs.cpp:155-222 with VectorType=float vec,
// Scalar=double (deduced from the double& invariants at the call site :237-244)
// ----------------------------------------------------------------------------------------------
double seg_dist_inv(V3 p1, V3 p2, V3 q1, V3 q2, double& invariant1, double& invariant2) {
    const double kSmallNumber = 0.0001;
    V3 u = p2 - p1;
    V3 v = q2 - q1;
    V3 w = p1 - q1;
    double a = dot(u, u);
    double b = dot(u, v);
    double c = dot(v, v);
    double d = dot(u, w);
    double e = dot(v, w);
    double f = a * c - b * b;
    double s1 = 0.0, s2 = f, t1 = 0.0, t2 = f;
    if (f < kSmallNumber) {
        s1 = 0.0; s2 = 1.0; t1 = e; t2 = c;
    } else {
        s1 = (b * e - c * d);
        t1 = (a * e - b * d);
        if (s1 < 0.0) { s1 = 0.0; t1 = e; t2 = c; }
        else if (s1 > s2) { s1 = s2; t1 = e + b; t2 = c; }
    }
    if (t1 < 0.0) {
        t1 = 0.0;
        if (-d < 0.0) s1 = 0.0;
        else if (-d > a) s1 = s2;
        else { s1 = -d; s2 = a; }
    } else if (t1 > t2) {
        t1 = t2;
        if ((-d + b) < 0.0) s1 = 0;
        else if ((-d + b) > a) s1 = s2;
        else { s1 = (-d + b); s2 = a; }
    }
    invariant1 = (fabs(s1) < kSmallNumber ? 0.0 : s1 / s2);
    invariant2 = (fabs(t1) < kSmallNumber ? 0.0 : t1 / t2);
    // double * Vector3f: the scalar is narrowed to the vector's scalar type (float)
    float i1 = (float)invariant1, i2 = (float)invariant2;
    return (double)norm((w + (i1 * u)) - (i2 * v));
}

// acosf-derived predicate -- stocs.cpp:1028-1032:
//   float angle_n = std::acos(dot)*180/M_PI;  if (angle_n < 30) ...
// std::acos(float) is acosf; acosf(x)*180 is a float product; "/M_PI" promotes to double; the
// result is narrowed to float.  NaN (dot > 1) compares false (Q7).
inline bool normal_compatible(float d) {
    float angle_n = (float)((double)(acosf(d) * 180) / M_PI);
    return angle_n < 30;
}
// internal-angle predicate -- stocs.cpp:428-429,440 (unqualified acos -> double):
//   float int_angle = acos(dot)*180/M_PI; int_angle = std::min(int_angle, 180-int_angle);
//   reject if int_angle < threshold
inline bool internal_angle_reject(float d, float threshold) {
    float int_angle = (float)(acos((double)d) * 180 / M_PI);
    float other = 180 - int_angle;
    float m = (other < int_angle) ? other : int_angle;  // std::min(a,b) = (b<a)?b:a
    return m < threshold;
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// the estimator state: include/stocs.hpp:151-177
// ----------------------------------------------------------------------------------------------
struct orc_ctx {
    orc_params prm;
    std::vector<Pt> scene, model;
    V3 centroid_scene, centroid_model;
    KdTree kd;
    orc_index* index;
    std::vector<uint8_t> edge_map;           // png values; all zero == "file absent" image of zeros
    std::vector<int> last_segment;           // indices of the points pushed to `segment` by the last sample_instance_base
    std::vector<uint8_t> previous_segment;   // cv::Mat previous_segment
    std::vector<uint8_t> segmentation_buffer;
    std::map<int, std::vector<uint8_t> > seg_masks;  // in-memory stand-in for dbg/seg_mask_<n>.png (Q14)
    // per-run state (stocs.hpp:166-167)
    std::vector<std::array<float, 16> > all_transforms, all_pose;
    std::vector<int> all_base_index;
    // model normalisation cache (recomputed per base in the reference, stocs.cpp:759-760)
    std::vector<V3> unit_pts;
    V3 gcenter;
    float ratio;
};

namespace {

// row 13: centroid_shift -- stocs.cpp:943-964
void centroid_shift(orc_ctx* c) {
    V3 cs = mk(0, 0, 0), cm = mk(0, 0, 0);
    for (size_t i = 0; i < c->scene.size(); ++i) cs = cs + c->scene[i].pos;
    for (size_t i = 0; i < c->model.size(); ++i) cm = cm + c->model[i].pos;
    cs = cs / (float)c->scene.size();
    cm = cm / (float)c->model.size();
    for (size_t i = 0; i < c->scene.size(); ++i) c->scene[i].pos = c->scene[i].pos - cs;
    for (size_t i = 0; i < c->model.size(); ++i) c->model[i].pos = c->model[i].pos - cm;
    c->centroid_scene = cs;
    c->centroid_model = cm;
}

// row 9: PairCreationFunctor::synch3DContent -- pairCreationFunctor.h:96-132; worldToUnit :71-75
void synch3DContent(orc_ctx* c) {
    AABB bbox;
    const size_t n = c->model.size();
    c->unit_pts.resize(n);
    for (size_t i = 0; i < n; ++i) { c->unit_pts[i] = c->model[i].pos; bbox.extendTo(c->model[i].pos); }
    c->gcenter = bbox.center();
    V3 ext = bbox.mx - bbox.mn;
    // depth()=(max-min)(2), width()=(1), height()=(0); float + 0.001 (double), max in double, stored float
    double r = std::max((double)ext.z + 0.001, std::max((double)ext.y + 0.001, (double)ext.x + 0.001));
    c->ratio = (float)r;
    const V3 half = mk(0.5f, 0.5f, 0.5f);
    for (size_t i = 0; i < n; ++i) c->unit_pts[i] = (c->unit_pts[i] - c->gcenter) / c->ratio + half;
}

void ppf_of(const Pt& a, const Pt& b, const orc_params& prm, int* out) {
    ppf_compute(a.pos, a.nrm, b.pos, b.nrm, prm.ppf_tr_discretization, prm.ppf_rot_discretization, 0, out);
}

// row 7: try_sampled_base -- stocs.cpp:224-268
bool try_sampled_base(const V3 base[4], float& invariant1, float& invariant2, int ids[4]) {
    float min_distance = std::numeric_limits<float>::max();
    int best1 = -1, best2 = -1, best3 = -1, best4 = -1;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) {
            if (i == j) continue;
            int k = 0;
            while (k == i || k == j) k++;
            int l = 0;
            while (l == i || l == j || l == k) l++;
            double li1, li2;
            float segment_distance = (float)seg_dist_inv(base[i], base[j], base[k], base[l], li1, li2);
            if (segment_distance < min_distance) {
                min_distance = segment_distance;
                best1 = i; best2 = j; best3 = k; best4 = l;
                invariant1 = (float)li1;
                invariant2 = (float)li2;
            }
        }
    }
    if (best1 < 0 || best2 < 0 || best3 < 0 || best4 < 0) return false;
    int tmp[4] = {ids[0], ids[1], ids[2], ids[3]};
    ids[0] = tmp[best1]; ids[1] = tmp[best2]; ids[2] = tmp[best3]; ids[3] = tmp[best4];
    return true;
}

// pass bodies of sample_class_base / sample_instance_base -- stocs.cpp:395-407, 424-442, 456-497
// (identical text at 597-608, 656-674, 688-729).  Sets w[i]=0 where the reference calls
// update_probability(0); other entries keep their value.
void pass1(orc_ctx* c, int b1, float* w) {
    const int n = (int)c->scene.size();
    for (int i = 0; i < n; ++i) {
        int ppf[4];
        ppf_of(c->scene[b1], c->scene[i], c->prm, ppf);
        if (!index_exists(c->index, ppf) || i == b1) w[i] = 0;
    }
}
void pass2(orc_ctx* c, int b1, int b2, float* w) {
    const int n = (int)c->scene.size();
    V3 v_1 = normalized(c->scene[b2].pos - c->scene[b1].pos);
    for (int i = 0; i < n; ++i) {
        V3 v_2 = normalized(c->scene[i].pos - c->scene[b1].pos);
        bool ang = internal_angle_reject(dot(v_1, v_2), c->prm.internal_angle_threshold);
        int ppf[4];
        ppf_of(c->scene[b2], c->scene[i], c->prm, ppf);
        if (!index_exists(c->index, ppf) || i == b2 || ang) w[i] = 0;
    }
}
void pass3(orc_ctx* c, int b1, int b2, int b3, float* w) {
    const int n = (int)c->scene.size();
    double x1 = c->scene[b1].pos.x, y1 = c->scene[b1].pos.y, z1 = c->scene[b1].pos.z;
    double x2 = c->scene[b2].pos.x, y2 = c->scene[b2].pos.y, z2 = c->scene[b2].pos.z;
    double x3 = c->scene[b3].pos.x, y3 = c->scene[b3].pos.y, z3 = c->scene[b3].pos.z;
    float denom = (float)(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
    float A = 0, B = 0, C = 0;
    if (denom != 0) {
        A = (float)((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / denom);
        B = (float)((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / denom);
        C = (float)((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / denom);
    }
    for (int i = 0; i < n; ++i) {
        float planar_distance = 10000;
        if (denom != 0) {
            const V3 p = c->scene[i].pos;
            planar_distance = (float)fabs((double)(A * p.x + B * p.y + C * p.z) - 1.0);
        }
        int ppf[4];
        ppf_of(c->scene[b3], c->scene[i], c->prm, ppf);
        const V3 pi = c->scene[i].pos;
        if (planar_distance > c->prm.plane_threshold ||
            norm(pi - c->scene[b1].pos) < c->prm.min_distance_base ||
            norm(pi - c->scene[b2].pos) < c->prm.min_distance_base ||
            norm(pi - c->scene[b3].pos) < c->prm.min_distance_base ||
            !index_exists(c->index, ppf) || i == b3)
            w[i] = 0;
    }
}

int draw_scene(orc_ctx* c, uint64_t seed, uint64_t attempt, uint64_t k) {
    std::vector<float> w(c->scene.size());
    for (size_t i = 0; i < w.size(); ++i) w[i] = c->scene[i].cur_prob;
    return draw_index(w.data(), (int)w.size(), rng64(seed, attempt, k));
}
void pull_w(orc_ctx* c, std::vector<float>& w) {
    w.resize(c->scene.size());
    for (size_t i = 0; i < w.size(); ++i) w[i] = c->scene[i].cur_prob;
}
void push_w(orc_ctx* c, const std::vector<float>& w) {
    for (size_t i = 0; i < w.size(); ++i) c->scene[i].cur_prob = w[i];
}

// rows 10: IndexedNormalSet<Point,3,7,float> -- normalset.h:65-151, normalset.hpp:57-214
struct NormalSet {
    static constexpr int NG = 7;
    float nepsilon;   // normalset.h:86
    float epsilon;    // cell edge in unit cube
    int egSize;
    std::unordered_map<int, std::array<std::vector<unsigned>, 343> > grid;  // lazily allocated AngularGrid

    explicit NormalSet(float eps) {
        nepsilon = (float)((double)(1.0f / (float)NG) + 0.00001);
        const int gridDepth = (int)(-log2f(eps));       // normalset.h:117 (std::log2(float))
        egSize = (int)pow(2.0, (double)gridDepth);      // :118
        epsilon = 1.f / egSize;                          // :119
    }
    // utils.h:139-148 UnrollIndexLoop (no validation in release): IndexT(coord[d]) truncation
    int indexPos(V3 p) const {
        V3 cp = p / epsilon;
        return (int)cp.z * egSize * egSize + ((int)cp.y * egSize + (int)cp.x);
    }
    int indexNormal(V3 n) const {
        const V3 half = mk(0.5f, 0.5f, 0.5f);
        V3 cn = (n / 2.0f + half) / nepsilon;
        return (int)cn.z * NG * NG + ((int)cn.y * NG + (int)cn.x);
    }
    void addElement(V3 p, V3 n, unsigned id) {  // normalset.hpp:114-131
        const int pId = indexPos(p);
        const int nId = indexNormal(n);
        if (nId < 0 || nId >= 343) return;  // .at() would throw (NaN direction)
        grid[pId][nId].push_back(id);
    }
    static int nbSample(float cosAlpha, float* angleStepOut, float* alphaOut) {
        const float alpha = acosf(cosAlpha);                                  // :178
        const float perimeter = (float)((double)2.0f * M_PI * (double)atanf(alpha));  // :179 (sic, Q10)
        const unsigned nb = (unsigned)(2 * ceilf(perimeter * (float)NG / 2.0f));      // :180
        if (angleStepOut) *angleStepOut = (float)((double)2.0f * M_PI / (double)(float)nb);  // :181
        if (alphaOut) *alphaOut = alpha;
        return (int)nb;
    }
    // Eigen Quaternion::setFromTwoVectors(z, n) + operator*(vec): see DESIGN.md
    static void quat_from_z(V3 n, float q[4] /*x,y,z,w*/) {
        V3 v0 = normalized(mk(0.f, 0.f, 1.f));
        V3 v1 = normalized(n);
        float c = dot(v1, v0);
        if (c < -1.0f + 1e-5f) {
            // Eigen falls back to an SVD here: the axis is the right-singular vector of [v0; v1] for the vanishing singular value,
            // i.e. the unit vector orthogonal to both = normalize(v0 x v1) (signed so that the rotation takes v0 to v1); (1,0,0) when
            // they are exactly opposite.  The sign and the degenerate choice of the library's own SVD are unpinned (Eigen absent);
            // probability ~1e-5 per query normal
            c = std::max(c, -1.0f);
            float w2 = (1.0f + c) * 0.5f;
            float s = sqrtf(1.0f - w2);
            V3 x = cross(v0, v1);
            float x2 = dot(x, x);
            V3 ax = x2 > 0.0f ? x / sqrtf(x2) : mk(1.0f, 0.0f, 0.0f);
            q[0] = ax.x * s; q[1] = ax.y * s; q[2] = ax.z * s; q[3] = sqrtf(w2);
            return;
        }
        V3 axis = cross(v0, v1);
        float s = sqrtf((1.0f + c) * 2.0f);
        float invs = 1.0f / s;
        q[0] = axis.x * invs; q[1] = axis.y * invs; q[2] = axis.z * invs; q[3] = s * 0.5f;
    }
    static V3 quat_rot(const float q[4], V3 v) {
        V3 qv = mk(q[0], q[1], q[2]);
        V3 uv = cross(qv, v);
        uv = uv + uv;
        return (v + q[3] * uv) + cross(qv, uv);
    }
    void getNeighbors(V3 p, V3 n, float cosAlpha, std::vector<unsigned>& nei) const {  // :166-214
        auto git = grid.find(indexPos(p));   // only the query's own position cell (Q9)
        if (git == grid.end()) return;
        float angleStep, alpha;
        const int nb = nbSample(cosAlpha, &angleStep, &alpha);
        const float sinAlpha = sinf(alpha);
        float q[4];
        quat_from_z(n, q);
        std::set<unsigned> colored;
        for (int a = 0; a != nb; a++) {
            float theta = (float)a * angleStep;
            V3 dir = normalized(quat_rot(q, mk(sinAlpha * cosf(theta), sinAlpha * sinf(theta), cosAlpha)));
            int id = indexNormal(dir);
            if (id < 0 || id >= 343) continue;
            if (git->second[id].size() != 0) colored.insert((unsigned)id);
        }
        for (std::set<unsigned>::const_iterator it = colored.begin(); it != colored.end(); ++it) {
            const std::vector<unsigned>& l = git->second[*it];
            nei.insert(nei.end(), l.begin(), l.end());
        }
    }
};

// row 8: find_congruent_sets_on_model -- stocs.cpp:753-869
// `seq` (optional) receives the same quads in WALK order: sorted by (position cell of the Q pair's query point, index
// position of the Q pair, index position of the P pair), the index position of a model pair being (its own quantised
// feature F, then (id1, id2)).  This enumeration exists only for the seeded subset rule of orc_run (divergence Q5: the
// reference shuffles with the C library's unseeded generator, so any fixed enumeration serves); it is the order in
// which the product's sort-based join meets the quads, so the product can resolve a drawn rank without writing them.
bool find_congruent(orc_ctx* c, const int ids[4], float invariant1, float invariant2,
                    std::vector<std::array<int, 4> >* quads, std::vector<std::array<int, 4> >* seq = NULL) {
    const Pt* B[4] = {&c->scene[ids[0]], &c->scene[ids[1]], &c->scene[ids[2]], &c->scene[ids[3]]};
    int ppf_1[4], ppf_2[4];
    ppf_of(*B[0], *B[1], c->prm, ppf_1);
    ppf_of(*B[2], *B[3], c->prm, ppf_2);
    std::vector<IPair> P_pairs, Q_pairs;
    std::vector<std::array<int, 4> > P_feat, Q_feat;
    index_lookup(c->index, ppf_1, &P_pairs, seq ? &P_feat : NULL);
    index_lookup(c->index, ppf_2, &Q_pairs, seq ? &Q_feat : NULL);
    quads->clear();  // (the reference clears after the early return; callers start empty anyway)
    if (seq) seq->clear();
    if (P_pairs.size() == 0 || Q_pairs.size() == 0) return false;

    const float alpha = dot(normalized(B[1]->pos - B[0]->pos), normalized(B[3]->pos - B[2]->pos));
    const float eps = c->prm.distance_threshold / c->ratio;  // getNormalizedEpsilon :141-143
    NormalSet nset(eps);
    for (size_t i = 0; i < P_pairs.size(); ++i) {
        const V3 p1 = c->unit_pts[P_pairs[i].first];
        const V3 p2 = c->unit_pts[P_pairs[i].second];
        const V3 n = normalized(p2 - p1);
        nset.addElement(p1 + invariant1 * (p2 - p1), n, (unsigned)i);
    }
    std::set<std::pair<unsigned, unsigned> > comb;
    std::vector<unsigned> nei;
    typedef std::array<int, 13> WalkKey;   // query cell | Q feature, Q pair | P feature, P pair
    std::vector<std::pair<WalkKey, std::array<int, 4> > > walk;
    for (unsigned i = 0; i < Q_pairs.size(); ++i) {
        const V3 p1 = c->unit_pts[Q_pairs[i].first];
        const V3 p2 = c->unit_pts[Q_pairs[i].second];
        const V3 pq1 = c->model[Q_pairs[i].first].pos;
        const V3 pq2 = c->model[Q_pairs[i].second].pos;
        nei.clear();
        const V3 query = p1 + invariant2 * (p2 - p1);
        const V3 queryQ = pq1 + invariant2 * (pq2 - pq1);
        const V3 queryn = normalized(p2 - p1);
        nset.getNeighbors(query, queryn, alpha, nei);
        for (unsigned k = 0; k != nei.size(); k++) {
            const int id = (int)nei[k];
            const V3 pp1 = c->model[P_pairs[id].first].pos;
            const V3 pp2 = c->model[P_pairs[id].second].pos;
            const V3 invPoint = pp1 + (pp2 - pp1) * invariant1;
            // squared metres compared with metres (Q1), reproduced
            if (sqn(queryQ - invPoint) <= c->prm.distance_threshold) {
                const bool fresh = comb.insert(std::make_pair((unsigned)id, i)).second;
                if (seq && fresh) {
                    std::array<int, 4> q = {{P_pairs[id].first, P_pairs[id].second, Q_pairs[i].first, Q_pairs[i].second}};
                    WalkKey k = {{nset.indexPos(query), Q_feat[i][0], Q_feat[i][1], Q_feat[i][2], Q_feat[i][3], Q_pairs[i].first, Q_pairs[i].second,
                                  P_feat[id][0], P_feat[id][1], P_feat[id][2], P_feat[id][3], P_pairs[id].first, P_pairs[id].second}};
                    walk.push_back(std::make_pair(k, q));
                }
            }
        }
    }
    if (seq) {
        std::sort(walk.begin(), walk.end());
        for (size_t k = 0; k < walk.size(); ++k) seq->push_back(walk[k].second);
    }
    for (auto it = comb.begin(); it != comb.end(); ++it) {
        std::array<int, 4> q = {{P_pairs[it->first].first, P_pairs[it->first].second,
                                 Q_pairs[it->second].first, Q_pairs[it->second].second}};
        quads->push_back(q);
    }
    return quads->size() != 0;
}

// rows 11/12: ComputeRigidTransformation stocs.cpp:270-361 +
// get_rigid_transform_from_congruent_pair stocs.cpp:871-941
bool rigid_transform(orc_ctx* c, const int ids[4], const int quad[4], float* T, float* pose) {
    const V3 p0 = c->scene[ids[0]].pos, p1 = c->scene[ids[1]].pos, p2 = c->scene[ids[2]].pos;
    const V3 q0 = c->model[quad[0]].pos, q1 = c->model[quad[1]].pos, q2 = c->model[quad[2]].pos;
    const V3 centroid1 = ((p0 + p1) + p2) / 3.0f;   // :885
    const V3 centroid2 = ((q0 + q1) + q2) / 3.0f;   // :907-909

    // degenerate frames: the reference returns kLargeNumber from a bool function (=> true with an
    // uninitialised matrix, Q2).  Deliberate divergence: reject.
    V3 vp1 = p1 - p0;
    if (sqn(vp1) == 0) return false;
    vp1 = normalized(vp1);
    V3 vp2 = (p2 - p0) - (dot(p2 - p0, vp1) * vp1);
    if (sqn(vp2) == 0) return false;
    vp2 = normalized(vp2);
    V3 vp3 = cross(vp1, vp2);
    V3 vq1 = q1 - q0;
    if (sqn(vq1) == 0) return false;
    vq1 = normalized(vq1);
    V3 vq2 = (q2 - q0) - (dot(q2 - q0, vq1) * vq1);
    if (sqn(vq2) == 0) return false;
    vq2 = normalized(vq2);
    V3 vq3 = cross(vq1, vq2);

    // rotation = rotate_p.transpose() * rotate_q, frames as rows (:316-326)
    M3 Pt_, Q_;
    const V3 pr[3] = {vp1, vp2, vp3}, qr[3] = {vq1, vq2, vq3};
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col) {
            Pt_.m[col][r] = pr[r][col];  // transpose
            Q_.m[r][col] = qr[r][col];
        }
    M3 R = mul(Pt_, Q_);
    // (rotation*rotation).diagonal() - 1 > 1e-6 (sic: R*R, Q3) :329
    M3 RR = mul(R, R);
    const float kSmall = 1e-6f;
    if ((RR.m[0][0] - 1.0f > kSmall) || (RR.m[1][1] - 1.0f > kSmall) || (RR.m[2][2] - 1.0f > kSmall)) return false;
    // rms is >= 0 whenever finite; a NaN rms fails "rms >= 0" (:922)
    {
        float rms = 0.0f;
        const V3 qs[3] = {q0, q1, q2}, ps[3] = {p0, p1, p2};
        for (int i = 0; i < 3; ++i) {
            V3 first = 1.0f * qs[i] - centroid2;
            V3 transformed = mul(R, first);
            rms += norm((transformed - ps[i]) + centroid1);
        }
        rms /= 4.0f;
        if (!(rms >= 0.0f)) return false;
    }
    // etrans = I; scale(1); translate(c1); rotate(R); translate(-c2)  (:348-357)
    V3 t = centroid1 + mul(R, -centroid2);
    for (int i = 0; i < 16; ++i) T[i] = 0.0f;
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col) T[col * 4 + r] = R.m[r][col];
    T[12] = t.x; T[13] = t.y; T[14] = t.z; T[15] = 1.0f;
    // camera-frame copy (:925-933): col(3) = c1 + centroid_scene - rot*scale*(c2 + centroid_model);
    // rot*scale from computeRotationScaling is the linear part itself (A = U S V^T = (U V^T)(V S V^T)),
    // so the SVD is skipped (equal up to float rounding of the SVD; documented).
    for (int i = 0; i < 16; ++i) pose[i] = T[i];
    V3 tc = (centroid1 + c->centroid_scene) - mul(R, centroid2 + c->centroid_model);
    pose[12] = tc.x; pose[13] = tc.y; pose[14] = tc.z; pose[15] = 1.0f;
    return true;
}

// row 16: compute_alignment_score_for_rigid_transform -- stocs.cpp:1006-1041
// `exact` (optional): the same weights summed in double -- NOT what the reference computes; a checker for the checker:
// the product adds the weights as integers and returns the exact mean, so it must agree with this value to float
// rounding, while the reference's running float sum (the return value) drifts from it by ~1e-9 * |M| on big models.
float lcp_score(const orc_ctx* c, const float* T, int32_t* hit, uint8_t* counted, double* exact = NULL) {
    const float epsilon = c->prm.distance_threshold;
    float weighted_match = 0;
    double weighted_exact = 0;
    const int n = (int)c->model.size();
    const float sq_eps = epsilon * epsilon;
    for (int i = 0; i < n; ++i) {
        int resId = c->kd.queryRestrictedClosestIndex(xform_point(T, c->model[i].pos), sq_eps);
        if (hit) hit[i] = resId;
        if (counted) counted[i] = 0;
        if (resId != -1) {
            V3 n_q = xform_normal(T, c->model[i].nrm);
            if (normal_compatible(dot(c->scene[resId].nrm, n_q))) {
                weighted_match += c->scene[resId].class_prob;
                weighted_exact += (double)c->scene[resId].class_prob;
                if (counted) counted[i] = 1;
            }
        }
    }
    if (exact) *exact = weighted_exact / (double)n;
    return weighted_match / (float)n;
}

// row 18 helpers: pose_clustering.cpp:5-25, 27-71
void quaternion_to_euler(const float q[4] /*x,y,z,w*/, float e[3]) {
    double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    // note: the reference multiplies floats (q.w()*q.x() ...) and then promotes; reproduce
    double sinr = +2.0 * (double)(q[3] * q[0] + q[1] * q[2]);
    double cosr = +1.0 - 2.0 * (double)(q[0] * q[0] + q[1] * q[1]);
    e[0] = (float)atan2(sinr, cosr);
    double sinp = +2.0 * (double)(q[3] * q[1] - q[2] * q[0]);
    if (fabs(sinp) >= 1) e[1] = (float)copysign(M_PI / 2, sinp);
    else e[1] = (float)asin(sinp);
    double siny = +2.0 * (double)(q[3] * q[2] + q[0] * q[1]);
    double cosy = +1.0 - 2.0 * (double)(q[1] * q[1] + q[2] * q[2]);
    e[2] = (float)atan2(siny, cosy);
    (void)qx; (void)qy; (void)qz; (void)qw;
}
// Eigen Quaternion(Matrix3) (Shoemake), as in Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl
void mat_to_quat(const M3& a, float q[4]) {
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
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (a.m[k][j] - a.m[j][k]) * t;
        q[j] = (a.m[j][i] + a.m[i][j]) * t;
        q[k] = (a.m[k][i] + a.m[i][k]) * t;
    }
}
M3 inverse3(const M3& a) {  // cofactor inverse (Eigen's 3x3 path)
    M3 r;
    float c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
    float c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
    float c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    float det = a.m[0][0] * c00 + (a.m[0][1] * c01 + a.m[0][2] * c02);
    float inv = 1.0f / det;
    r.m[0][0] = c00 * inv;
    r.m[1][0] = c01 * inv;
    r.m[2][0] = c02 * inv;
    r.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * inv;
    r.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * inv;
    r.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * inv;
    r.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * inv;
    r.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * inv;
    r.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * inv;
    return r;
}
void pose_diff(const float* test, const float* base, const float* sym, float& rot_err, float& tr_err) {
    M3 tr_, br_;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { tr_.m[i][j] = T_at(test, i, j); br_.m[i][j] = T_at(base, i, j); }
    M3 diff = mul(inverse3(tr_), br_);
    float q[4], e[3];
    mat_to_quat(diff, q);
    quaternion_to_euler(q, e);
    for (int d = 0; d < 3; ++d) {
        e[d] = (float)((double)e[d] * 180.0 / M_PI);
        e[d] = fabsf(e[d]);
        if (sym[d] == 90) {
            // `abs(float - 90)` picks the int overload of ::abs in the reference when only <cstdlib>
            // is visible; with <cmath> it is the float overload.  Float overload assumed.
            e[d] = fabsf(e[d] - 90);
            e[d] = std::min(e[d], 90 - e[d]);
        } else if (sym[d] == 180) {
            e[d] = std::min(e[d], 180 - e[d]);
        } else if (sym[d] == 360) {
            e[d] = 0;
        }
    }
    rot_err = std::max(std::max(e[0], e[1]), e[2]);
    tr_err = (float)sqrt(pow((double)(T_at(base, 0, 3) - T_at(test, 0, 3)), 2) +
                         pow((double)(T_at(base, 1, 3) - T_at(test, 1, 3)), 2) +
                         pow((double)(T_at(base, 2, 3) - T_at(test, 2, 3)), 2));
}

// row 5 helper: generate_segmentation_mask -- rgbd.cpp:314-367 (PNG round trip replaced by seg_masks)
void generate_segmentation_mask(orc_ctx* c, int prow, int pcol, float max_distance,
                                std::vector<uint8_t>& closed_list, int base_num) {
    const int W = c->prm.image_width, H = c->prm.image_height;
    int segment_index = c->segmentation_buffer[(size_t)prow * W + pcol];
    if (segment_index != 0) {
        closed_list = c->seg_masks[segment_index];
        return;
    }
    std::queue<std::pair<int, int> > open_list;
    open_list.push(std::make_pair(prow, pcol));
    while (!open_list.empty()) {
        std::pair<int, int> curr = open_list.front();
        closed_list[(size_t)curr.first * W + curr.second] = 255;
        c->segmentation_buffer[(size_t)curr.first * W + curr.second] = (uint8_t)base_num;
        open_list.pop();
        for (int i = curr.first - 1; i <= curr.first + 1; i += 1) {
            for (int j = curr.second - 1; j <= curr.second + 1; j += 1) {
                if (i < 0 || j < 0 || i >= H || j >= W) continue;
                float edge_probability = (float)(255.0 - c->edge_map[(size_t)i * W + j]) / 255.0;
                int expanded = (int)closed_list[(size_t)i * W + j];
                float dist = (float)sqrt(pow((double)(prow - i), 2) + pow((double)(pcol - j), 2));
                if (expanded == 0 && edge_probability == 0 && dist < max_distance) {
                    open_list.push(std::make_pair(i, j));
                    closed_list[(size_t)i * W + j] = 255;
                    c->segmentation_buffer[(size_t)i * W + j] = (uint8_t)base_num;
                }
            }
        }
    }
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

void orc_default_params(orc_params* p) {
    p->distance_threshold = 0.005f;
    p->ppf_tr_discretization = 5;
    p->ppf_rot_discretization = 5;
    p->plane_threshold = 0.015f;
    p->min_distance_base = 0.01f;
    p->internal_angle_threshold = 30;
    p->image_width = 640;
    p->image_height = 480;
}

int orc_ppf_closest_bin(int value, int discretization) { return closest_bin(value, discretization); }

void orc_ppf_compute(const float* p1, const float* n1, const float* p2, const float* n2, int tr, int rot,
                     int mode, int* out4) {
    ppf_compute(ld(p1), ld(n1), ld(p2), ld(n2), tr, rot, mode, out4);
}

// Threads for the feature evaluation of orc_index_build (default 1).  The insertion stays sequential in the reference's loop order, so
// the index is the same whatever the count; bench.py raises it to the cgroup's CPU share so that the CPU whole-path baseline does not
// spend ten seconds of the bench run on the (offline, untimed) 25 M-pair index of the metric model.
static int g_index_build_threads = 1;
void orc_set_index_build_threads(int n) { g_index_build_threads = n > 1 ? n : 1; }

orc_index* orc_index_build(const float* pos3, const float* nrm3, int n, int tr, int rot) {
    orc_index* ix = new orc_index;
    ix->tr = tr; ix->rot = rot; ix->npairs = 0;
    const bool dbg_t = getenv("ORC_DEBUG_TIMING") != NULL;
    double t_feat = 0, t_ins = 0; const double t_begin = omp_get_wtime();
    // stocs.cpp:63-78: ordered pairs, id1 outer, id2 inner, id1 != id2 -- the features of a block of id1 rows are evaluated side by
    // side, then appended to their feature's list in exactly that order.  While the features stay inside a small box (they do: a
    // distance bin and three angle bins) the list of a feature is found through a dense table instead of a hash look-up per pair;
    // the map the look-ups use is filled from the lists at the end -- same content, same order inside every list.
    const int ROWS = 64;
    std::vector<std::array<int, 4> > feat((size_t)ROWS * (size_t)(n > 0 ? n : 1));
    const int NA = 180 / (rot > 0 ? rot : 1) + 3, ND = 4096;                 // dense box: angles 0 .. 180 + 2 rot, distance bins 0 .. 4095
    const bool dense_ok = tr > 0 && rot > 0 && (size_t)NA * NA * NA * 64 <= ((size_t)1 << 31);
    std::vector<int32_t> slot;                                               // (grown by distance bin) dense cell -> list id + 1
    std::vector<std::array<int, 4> > list_key;
    std::vector<std::vector<IPair> > lists;
    const size_t per_d = (size_t)NA * NA * NA;
    for (int r0 = 0; r0 < n; r0 += ROWS) {
        const int r1 = std::min(n, r0 + ROWS);
        const double ta = omp_get_wtime();
#pragma omp parallel for num_threads(g_index_build_threads) schedule(static)
        for (int id1 = r0; id1 < r1; ++id1)
            for (int id2 = 0; id2 < n; ++id2) {
                if (id1 == id2) continue;
                int f[4];
                ppf_compute(ld(pos3 + 3 * id1), ld(nrm3 + 3 * id1), ld(pos3 + 3 * id2), ld(nrm3 + 3 * id2), tr, rot, 0, f);
                feat[(size_t)(id1 - r0) * n + id2] = {{f[0], f[1], f[2], f[3]}};
            }
        const double tb = omp_get_wtime(); t_feat += tb - ta;
        for (int id1 = r0; id1 < r1; ++id1)
            for (int id2 = 0; id2 < n; ++id2) {
                if (id1 == id2) continue;
                const std::array<int, 4>& F = feat[(size_t)(id1 - r0) * n + id2];
                bool in_box = dense_ok && F[0] >= 0 && F[0] % tr == 0 && F[0] / tr < ND;
                for (int k = 1; k < 4 && in_box; ++k) in_box = F[k] >= 0 && F[k] % rot == 0 && F[k] / rot < NA;
                if (in_box) {
                    const size_t cell = (((size_t)(F[0] / tr) * NA + (size_t)(F[1] / rot)) * NA + (size_t)(F[2] / rot)) * NA + (size_t)(F[3] / rot);
                    if (cell >= slot.size()) slot.resize(((size_t)(F[0] / tr) + 8) * per_d, 0);
                    int32_t& sl = slot[cell];
                    if (!sl) { lists.push_back(std::vector<IPair>()); list_key.push_back(F); sl = (int32_t)lists.size(); }
                    lists[(size_t)sl - 1].push_back(IPair(id1, id2));
                } else {
                    ix->base[F].push_back(IPair(id1, id2));
                }
                ix->npairs++;
            }
        t_ins += omp_get_wtime() - tb;
    }
    for (size_t l = 0; l < lists.size(); ++l) {     // a feature is either inside the box or not: no list exists twice
        std::vector<IPair>& dst = ix->base[list_key[l]];
        dst.swap(lists[l]);
    }
    if (dbg_t) fprintf(stderr, "[oracle index] features %.2f s (%d threads), insertion %.2f s, %zu distinct features\n", t_feat, g_index_build_threads, t_ins, ix->base.size());
    // the key set of the reference's map: every stored feature under its 128 insertion offsets (rgbd.cpp:130-137)
    for (auto it = ix->base.begin(); it != ix->base.end(); ++it) {
        const std::array<int, 4>& F = it->first;
        for (int p1 = F[0] - tr; p1 < F[0] + tr; p1 += tr)
            for (int p2 = F[1] - 2 * rot; p2 < F[1] + 2 * rot; p2 += rot)
                for (int p3 = F[2] - 2 * rot; p3 < F[2] + 2 * rot; p3 += rot)
                    for (int p4 = F[3] - 2 * rot; p4 < F[3] + 2 * rot; p4 += rot) {
                        if (p1 <= 5 || p2 < 0 || p3 < 0 || p4 < 0) continue;
                        std::array<int, 4> key = {{p1, p2, p3, p4}};
                        ix->keys[key] = 1;
                    }
    }
    if (dbg_t) fprintf(stderr, "[oracle index] whole build %.2f s, %zu keys\n", omp_get_wtime() - t_begin, ix->keys.size());
    return ix;
}
void orc_index_free(orc_index* ix) { delete ix; }
int64_t orc_index_lookup(const orc_index* ix, const int* key4, int32_t* pairs2, int64_t cap) {
    std::vector<IPair> out;
    int64_t n = index_lookup(ix, key4, &out);
    for (int64_t i = 0; i < n && i < cap; ++i) { pairs2[2 * i] = out[i].first; pairs2[2 * i + 1] = out[i].second; }
    return n;
}
int orc_index_exists(const orc_index* ix, const int* key4) { return index_exists(ix, key4) ? 1 : 0; }
int64_t orc_index_num_pairs(const orc_index* ix) { return ix->npairs; }

orc_index_lit* orc_index_lit_build(const float* pos3, const float* nrm3, int n, int tr_i, int rot_i) {
    orc_index_lit* ix = new orc_index_lit;
    const float tr_discretization = (float)tr_i, rot_discretization = (float)rot_i;  // float params, rgbd.cpp:125-126
    for (int id1 = 0; id1 < n; ++id1)
        for (int id2 = 0; id2 < n; ++id2) {
            if (id1 == id2) continue;
            int f[4];
            ppf_compute(ld(pos3 + 3 * id1), ld(nrm3 + 3 * id1), ld(pos3 + 3 * id2), ld(nrm3 + 3 * id2), tr_i, rot_i, 0, f);
            // rgbd.cpp:130-153, loop variables int, bounds float
            for (int p1 = f[0] - tr_discretization; p1 < f[0] + tr_discretization; p1 += tr_discretization)
                for (int p2 = f[1] - 2 * rot_discretization; p2 < f[1] + 2 * rot_discretization; p2 += rot_discretization)
                    for (int p3 = f[2] - 2 * rot_discretization; p3 < f[2] + 2 * rot_discretization; p3 += rot_discretization)
                        for (int p4 = f[3] - 2 * rot_discretization; p4 < f[3] + 2 * rot_discretization; p4 += rot_discretization) {
                            if (p1 <= 5 || p2 < 0 || p3 < 0 || p4 < 0) continue;
                            std::vector<int> k = {p1, p2, p3, p4};
                            ix->map[k].push_back(IPair(id1, id2));
                        }
        }
    return ix;
}
void orc_index_lit_free(orc_index_lit* ix) { delete ix; }
int64_t orc_index_lit_lookup(const orc_index_lit* ix, const int* key4, int32_t* pairs2, int64_t cap) {
    std::vector<int> k = {key4[0], key4[1], key4[2], key4[3]};
    auto it = ix->map.find(k);
    if (it == ix->map.end()) return 0;
    int64_t n = (int64_t)it->second.size();
    for (int64_t i = 0; i < n && i < cap; ++i) { pairs2[2 * i] = it->second[i].first; pairs2[2 * i + 1] = it->second[i].second; }
    return n;
}
int64_t orc_index_lit_num_keys(const orc_index_lit* ix) { return (int64_t)ix->map.size(); }

orc_ctx* orc_ctx_create(const orc_params* prm, const float* sp, const float* sn, const float* sprob,
                        const int32_t* spix, int nS, const float* mp, const float* mn, int nM, int build_index) {
    orc_ctx* c = new orc_ctx;
    c->prm = *prm;
    c->scene.resize(nS);
    for (int i = 0; i < nS; ++i) {
        Pt& p = c->scene[i];
        p.pos = ld(sp + 3 * i);
        p.nrm = normalized(ld(sn + 3 * i));  // set_normal normalises, point3d.hpp:43-45
        p.row = spix ? spix[2 * i] : 0;
        p.col = spix ? spix[2 * i + 1] : 0;
        p.class_prob = sprob[i];
        p.cur_prob = sprob[i];
    }
    c->model.resize(nM);
    for (int i = 0; i < nM; ++i) {
        Pt& p = c->model[i];
        p.pos = ld(mp + 3 * i);
        p.nrm = normalized(ld(mn + 3 * i));
        p.row = p.col = 0;
        p.class_prob = p.cur_prob = 0;
    }
    // the PPF index is built from the model file as saved by pre_process_model, i.e. BEFORE the
    // estimator's centroid shift (stocs.cpp:59-78 vs 943-964); PPFs are translation invariant up to rounding
    c->index = NULL;
    if (build_index) {
        std::vector<float> nn((size_t)nM * 3);  // set_normal()-normalised normals
        for (int i = 0; i < nM; ++i) { nn[3 * i] = c->model[i].nrm.x; nn[3 * i + 1] = c->model[i].nrm.y; nn[3 * i + 2] = c->model[i].nrm.z; }
        c->index = orc_index_build(mp, nn.data(), nM, prm->ppf_tr_discretization, prm->ppf_rot_discretization);
    }
    centroid_shift(c);
    // kdtree_initialize -- stocs.cpp:966-980
    for (int i = 0; i < nS; ++i) c->kd.add(c->scene[i].pos);
    if (nS > 0) c->kd.finalize();
    synch3DContent(c);
    size_t px = (size_t)prm->image_width * prm->image_height;
    c->edge_map.assign(px, 0);
    c->previous_segment.assign(px, 0);
    c->segmentation_buffer.assign(px, 0);
    return c;
}
void orc_ctx_destroy(orc_ctx* c) {
    if (!c) return;
    if (c->index) orc_index_free(c->index);
    delete c;
}
void orc_get_centroids(const orc_ctx* c, float* s, float* m) {
    s[0] = c->centroid_scene.x; s[1] = c->centroid_scene.y; s[2] = c->centroid_scene.z;
    m[0] = c->centroid_model.x; m[1] = c->centroid_model.y; m[2] = c->centroid_model.z;
}
void orc_get_scene(const orc_ctx* c, float* pos3, float* prob, float* class_prob) {
    for (size_t i = 0; i < c->scene.size(); ++i) {
        if (pos3) { pos3[3 * i] = c->scene[i].pos.x; pos3[3 * i + 1] = c->scene[i].pos.y; pos3[3 * i + 2] = c->scene[i].pos.z; }
        if (prob) prob[i] = c->scene[i].cur_prob;
        if (class_prob) class_prob[i] = c->scene[i].class_prob;
    }
}
void orc_get_model(const orc_ctx* c, float* pos3) {
    for (size_t i = 0; i < c->model.size(); ++i) { pos3[3 * i] = c->model[i].pos.x; pos3[3 * i + 1] = c->model[i].pos.y; pos3[3 * i + 2] = c->model[i].pos.z; }
}
void orc_set_edge_map(orc_ctx* c, const uint8_t* edge) {
    size_t px = (size_t)c->prm.image_width * c->prm.image_height;
    c->edge_map.assign(edge, edge + px);
}
const orc_index* orc_ctx_index(const orc_ctx* c) { return c->index; }

uint64_t orc_rng(uint64_t seed, uint64_t attempt, uint64_t k) { return rng64(seed, attempt, k); }
int orc_draw(const float* w, int n, uint64_t r64) { return draw_index(w, n, r64); }

void orc_class_pass(orc_ctx* c, int pass, const int32_t* b3, const float* w_in, float* w_out) {
    const int n = (int)c->scene.size();
    if (w_in != w_out) memcpy(w_out, w_in, sizeof(float) * n);
    if (pass == 1) pass1(c, b3[0], w_out);
    else if (pass == 2) pass2(c, b3[0], b3[1], w_out);
    else pass3(c, b3[0], b3[1], b3[2], w_out);
}

// row 4: sample_class_base -- stocs.cpp:363-519
int orc_sample_class_base(orc_ctx* c, uint64_t seed, uint64_t attempt, int32_t* ids4, float* inv2) {
    const int n = (int)c->scene.size();
    if (n == 0) return 0;
    // :373-381: previous_segment is all zero in class mode; update_class_probability(1.0) is a no-op
    for (int i = 0; i < n; ++i) c->scene[i].cur_prob = c->scene[i].class_prob;
    std::vector<float> w;
    int b1 = draw_scene(c, seed, attempt, 0);
    if (b1 < 0 || c->scene[b1].cur_prob == 0.0f) { c->last_segment.clear(); return 0; }   // returns before `segment` is touched (:586-589)
    pull_w(c, w); pass1(c, b1, w.data()); push_w(c, w);
    int b2 = draw_scene(c, seed, attempt, 1);
    if (b2 < 0 || c->scene[b2].cur_prob == 0.0f) return 0;
    pull_w(c, w); pass2(c, b1, b2, w.data()); push_w(c, w);
    int b3 = draw_scene(c, seed, attempt, 2);
    if (b3 < 0 || c->scene[b3].cur_prob == 0.0f) return 0;
    pull_w(c, w); pass3(c, b1, b2, b3, w.data()); push_w(c, w);
    int b4 = draw_scene(c, seed, attempt, 3);
    if (b4 < 0 || c->scene[b4].cur_prob == 0.0f) return 0;
    int ids[4] = {b1, b2, b3, b4};
    V3 base[4] = {c->scene[b1].pos, c->scene[b2].pos, c->scene[b3].pos, c->scene[b4].pos};
    float i1 = 0, i2 = 0;
    bool ok = try_sampled_base(base, i1, i2, ids);
    for (int k = 0; k < 4; ++k) ids4[k] = ids[k];
    inv2[0] = i1; inv2[1] = i2;
    return ok ? 1 : 0;
}

// row 5: sample_instance_base -- stocs.cpp:559-751 (+ prune_edge_pixels :521-535)
// `segment` of the last orc_sample_instance_base (the out-parameter of stocs.cpp:559-565, filled at :628-638), as scene indices
int orc_get_segment(const orc_ctx* c, int32_t* idx, int cap) {
    for (size_t i = 0; i < c->last_segment.size() && (int)i < cap; ++i) idx[i] = c->last_segment[i];
    return (int)c->last_segment.size();
}
int orc_sample_instance_base(orc_ctx* c, uint64_t seed, uint64_t attempt, float dispersion, int base_num,
                             int32_t* ids4, float* inv2) {
    const int n = (int)c->scene.size();
    const int W = c->prm.image_width;
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i) {  // :572-580 (compounding decay)
        Pt& p = c->scene[i];
        int isPresent = (int)c->previous_segment[(size_t)p.row * W + p.col];
        if (isPresent) p.class_prob = dispersion * p.class_prob;
        p.cur_prob = p.class_prob;
    }
    for (int i = 0; i < n; ++i) {  // prune_edge_pixels
        Pt& p = c->scene[i];
        float edge_probability = (float)(255.0 - c->edge_map[(size_t)p.row * W + p.col]) / 255.0;
        if (edge_probability == 1) p.cur_prob = 0;
    }
    std::vector<float> w;
    int b1 = draw_scene(c, seed, attempt, 0);
    if (b1 < 0 || c->scene[b1].cur_prob == 0.0f) { c->last_segment.clear(); return 0; }   // returns before `segment` is touched (:586-589)
    pull_w(c, w); pass1(c, b1, w.data()); push_w(c, w);
    float max_pixel_distance = 0;  // :610-618
    for (int i = 0; i < n; ++i) {
        if (c->scene[i].cur_prob != 0) {
            float dist = (float)sqrt(pow((double)(c->scene[b1].row - c->scene[i].row), 2) +
                                     pow((double)(c->scene[b1].col - c->scene[i].col), 2));
            if (dist > max_pixel_distance) max_pixel_distance = dist;
        }
    }
    std::vector<uint8_t> segmentation_mask((size_t)W * c->prm.image_height, 0);
    generate_segmentation_mask(c, c->scene[b1].row, c->scene[b1].col, max_pixel_distance, segmentation_mask, base_num);
    c->seg_masks[base_num] = segmentation_mask;   // cv::imwrite(... seg_mask_<n>.png) :625
    c->previous_segment = segmentation_mask;      // :626
    c->last_segment.clear();
    for (int i = 0; i < n; ++i) {                 // :628-638: survivors inside the mask are pushed to `segment`, the others zeroed
        if (c->scene[i].cur_prob != 0) {
            int isValid = (int)segmentation_mask[(size_t)c->scene[i].row * W + c->scene[i].col];
            if (isValid) c->last_segment.push_back(i);
            else c->scene[i].cur_prob = 0;
        }
    }
    int b2 = draw_scene(c, seed, attempt, 1);
    if (b2 < 0 || c->scene[b2].cur_prob == 0.0f) return 0;
    pull_w(c, w); pass2(c, b1, b2, w.data()); push_w(c, w);
    int b3 = draw_scene(c, seed, attempt, 2);
    if (b3 < 0 || c->scene[b3].cur_prob == 0.0f) return 0;
    pull_w(c, w); pass3(c, b1, b2, b3, w.data()); push_w(c, w);
    int b4 = draw_scene(c, seed, attempt, 3);
    if (b4 < 0 || c->scene[b4].cur_prob == 0.0f) return 0;
    int ids[4] = {b1, b2, b3, b4};
    V3 base[4] = {c->scene[b1].pos, c->scene[b2].pos, c->scene[b3].pos, c->scene[b4].pos};
    float i1 = 0, i2 = 0;
    bool ok = try_sampled_base(base, i1, i2, ids);
    for (int k = 0; k < 4; ++k) ids4[k] = ids[k];
    inv2[0] = i1; inv2[1] = i2;
    return ok ? 1 : 0;
}

double orc_segment_distance_and_invariants(const float* p1, const float* p2, const float* q1, const float* q2,
                                           double* inv1, double* inv2) {
    return seg_dist_inv(ld(p1), ld(p2), ld(q1), ld(q2), *inv1, *inv2);
}
int orc_try_sampled_base(orc_ctx* c, int32_t* ids4, float* inv2) {
    int ids[4] = {ids4[0], ids4[1], ids4[2], ids4[3]};
    V3 base[4] = {c->scene[ids[0]].pos, c->scene[ids[1]].pos, c->scene[ids[2]].pos, c->scene[ids[3]].pos};
    float i1 = 0, i2 = 0;
    bool ok = try_sampled_base(base, i1, i2, ids);
    for (int k = 0; k < 4; ++k) ids4[k] = ids[k];
    inv2[0] = i1; inv2[1] = i2;
    return ok ? 1 : 0;
}

int64_t orc_find_congruent(orc_ctx* c, const int32_t* ids4, float inv1, float inv2, int32_t* quads4, int64_t cap) {
    int ids[4] = {ids4[0], ids4[1], ids4[2], ids4[3]};
    std::vector<std::array<int, 4> > quads;
    find_congruent(c, ids, inv1, inv2, &quads);
    for (size_t i = 0; i < quads.size() && (int64_t)i < cap; ++i)
        for (int k = 0; k < 4; ++k) quads4[4 * i + k] = quads[i][k];
    return (int64_t)quads.size();
}
int64_t orc_find_congruent_seq(orc_ctx* c, const int32_t* ids4, float inv1, float inv2, int32_t* quads4, int64_t cap) {
    int ids[4] = {ids4[0], ids4[1], ids4[2], ids4[3]};
    std::vector<std::array<int, 4> > quads, seq;
    find_congruent(c, ids, inv1, inv2, &quads, &seq);
    for (size_t i = 0; i < seq.size() && (int64_t)i < cap; ++i)
        for (int k = 0; k < 4; ++k) quads4[4 * i + k] = seq[i][k];
    return (int64_t)seq.size();
}
int orc_normalset_params(float eps_unit, int* gridDepth, int* egSize, float* cell) {
    NormalSet ns(eps_unit);
    if (gridDepth) *gridDepth = (int)(-log2f(eps_unit));
    if (egSize) *egSize = ns.egSize;
    if (cell) *cell = ns.epsilon;
    return 0;
}
int orc_cone_samples(float cos_alpha) { return NormalSet::nbSample(cos_alpha, NULL, NULL); }
int orc_index_normal(const float* n3) { NormalSet ns(0.03f); return ns.indexNormal(ld(n3)); }
float orc_model_ratio(orc_ctx* c, float* g) { if (g) { g[0] = c->gcenter.x; g[1] = c->gcenter.y; g[2] = c->gcenter.z; } return c->ratio; }

int orc_rigid_transform(orc_ctx* c, const int32_t* ids4, const int32_t* quad4, float* T, float* pose) {
    int ids[4] = {ids4[0], ids4[1], ids4[2], ids4[3]};
    int q[4] = {quad4[0], quad4[1], quad4[2], quad4[3]};
    return rigid_transform(c, ids, q, T, pose) ? 1 : 0;
}

int orc_nn(orc_ctx* c, const float* q3, float sqdist) { return c->kd.queryRestrictedClosestIndex(ld(q3), sqdist); }
int orc_nn_brute(orc_ctx* c, const float* q3, float sqdist, int* n_ties) {
    V3 q = ld(q3);
    int best = -1, ties = 0;
    float bd = sqdist;
    for (size_t i = 0; i < c->scene.size(); ++i) {
        float d = sqn(q - c->scene[i].pos);
        if (d <= sqdist) {
            if (best < 0 || d < bd) { best = (int)i; bd = d; ties = 0; }
            else if (d == bd) { ties++; best = (int)i; }
        }
    }
    if (n_ties) *n_ties = ties;
    return best;
}
float orc_lcp(orc_ctx* c, const float* T16) { return lcp_score(c, T16, NULL, NULL); }
void orc_lcp_batch(orc_ctx* c, const float* T16, int n, float* out, int nthreads) {
    (void)nthreads;
#ifdef _OPENMP
    if (nthreads > 1) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
        for (int i = 0; i < n; ++i) out[i] = lcp_score(c, T16 + 16 * (size_t)i, NULL, NULL);
        return;
    }
#endif
    for (int i = 0; i < n; ++i) out[i] = lcp_score(c, T16 + 16 * (size_t)i, NULL, NULL);
}
void orc_lcp_detail(orc_ctx* c, const float* T16, int32_t* hit, uint8_t* counted) { lcp_score(c, T16, hit, counted); }
// out: the reference's float-accumulated scores; out_exact: the same matches summed in double (see lcp_score)
void orc_lcp_batch_exact(orc_ctx* c, const float* T16, int n, float* out, double* out_exact, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 1 ? nthreads : 1) schedule(dynamic, 16)
    for (int i = 0; i < n; ++i) out[i] = lcp_score(c, T16 + 16 * (size_t)i, NULL, NULL, out_exact + i);
}
// row 17: compute_best_transform -- stocs.cpp:982-1004 (strict >, from 0: first max wins, Q18)
int orc_best(const float* lcp, int n, float* best_score) {
    float max_score = 0;
    int index = -1;
    for (int i = 0; i < n; ++i)
        if (lcp[i] > max_score) { max_score = lcp[i]; index = i; }
    if (best_score) *best_score = max_score;
    return index;
}
int orc_normal_compatible(float d) { return normal_compatible(d) ? 1 : 0; }
int orc_internal_angle_reject(float d, float threshold) { return internal_angle_reject(d, threshold) ? 1 : 0; }

void orc_pose_diff(const float* t, const float* b, const float* sym3, float* r, float* tr) { pose_diff(t, b, sym3, *r, *tr); }

// row 18: greedy_clustering -- pose_clustering.cpp:79-121.  std::sort is unstable in the reference;
// a stable sort by descending lcp is used here (ties keep input order; documented).
int orc_greedy_clustering(const float* poses16, const float* lcp, int n, float acceptable_fraction, float best_score,
                          int maximum_pose_count, float min_distance, float min_angle, const float* sym3,
                          int32_t* out_idx, int cap) {
    std::vector<int> pruned;
    for (int i = 0; i < n; ++i)
        if (lcp[i] > acceptable_fraction * best_score) pruned.push_back(i);
    std::stable_sort(pruned.begin(), pruned.end(), [&](int a, int b) { return lcp[a] > lcp[b]; });
    std::vector<int> clustered;
    for (size_t ci = 0; ci < pruned.size(); ++ci) {
        int cand = pruned[ci];
        bool inValid = false;
        for (size_t k = 0; k < clustered.size(); ++k) {
            float re, te;
            pose_diff(poses16 + 16 * (size_t)cand, poses16 + 16 * (size_t)clustered[k], sym3, re, te);
            if (re < min_angle && te < min_distance) { inValid = true; break; }
        }
        if (!inValid) clustered.push_back(cand);
        if ((int)clustered.size() > maximum_pose_count) break;
    }
    for (size_t i = 0; i < clustered.size() && (int)i < cap; ++i) out_idx[i] = clustered[i];
    return (int)clustered.size();
}

// row 19: run_stocs_estimation -- stocs_match_one_object.cpp:51-185, class mode, with the seeded
// divergences Q5 (subset choice) and Q6 (draws).  Subset rule when a base has >= max quads:
// partial Fisher-Yates over the base's quads in walk order (find_congruent's `seq`) with
// rng64(seed, 0x5E1EC7 + base_number, j); a base with fewer uses all of them in std::set order.
int orc_run_mode(orc_ctx* c, uint64_t seed, int number_of_bases, int maximum_congruent_sets, int instance_mode, float dispersion, orc_run_result* out);
int orc_run(orc_ctx* c, uint64_t seed, int number_of_bases, int maximum_congruent_sets, orc_run_result* out) {
    return orc_run_mode(c, seed, number_of_bases, maximum_congruent_sets, 0, 0.0f, out);
}
// instance_mode != 0: the caller's branch on the presence of probability_maps/edge.png (stocs_match_one_object.cpp:90):
// sample_instance_base(..., dispersion, i + 1) instead of sample_class_base
int orc_run_mode(orc_ctx* c, uint64_t seed, int number_of_bases, int maximum_congruent_sets, int instance_mode, float dispersion, orc_run_result* out) {
    typedef std::chrono::high_resolution_clock clk;
    struct Base { int ids[4]; float i1, i2; std::vector<std::array<int, 4> > quads, seq; };
    std::vector<Base> base_set;
    c->all_transforms.clear(); c->all_pose.clear(); c->all_base_index.clear();
    auto t0 = clk::now();
    for (int i = 0; i < number_of_bases; ++i) {
        int32_t ids[4] = {-1, -1, -1, -1};
        float inv[2];
        const int ok = instance_mode ? orc_sample_instance_base(c, seed, (uint64_t)i, dispersion, i + 1, ids, inv) : orc_sample_class_base(c, seed, (uint64_t)i, ids, inv);
        if (ok) {
            Base b;
            for (int k = 0; k < 4; ++k) b.ids[k] = ids[k];
            b.i1 = inv[0]; b.i2 = inv[1];
            base_set.push_back(b);
        }
    }
    auto t1 = clk::now();
    int total_quads = 0;
    for (size_t b = 0; b < base_set.size(); ++b) find_congruent(c, base_set[b].ids, base_set[b].i1, base_set[b].i2, &base_set[b].quads, &base_set[b].seq);
    for (size_t b = 0; b < base_set.size(); ++b) {
        const int nq = (int)base_set[b].quads.size();
        std::vector<int> pick;
        if (nq < maximum_congruent_sets) {
            for (int i = 0; i < nq; ++i) pick.push_back(i);
        } else {
            std::vector<int> perm(nq);
            for (int i = 0; i < nq; ++i) perm[i] = i;
            for (int j = 0; j < maximum_congruent_sets; ++j) {
                uint64_t r = rng64(seed, 0x5E1EC7ull + b, (uint64_t)j);
                int k = j + (int)(((unsigned __int128)r * (unsigned __int128)(uint64_t)(nq - j)) >> 64);
                std::swap(perm[j], perm[k]);
                pick.push_back(perm[j]);
            }
        }
        for (size_t i = 0; i < pick.size(); ++i) {
            std::array<float, 16> T, P;
            const std::array<int, 4>& qq = (nq < maximum_congruent_sets) ? base_set[b].quads[pick[i]] : base_set[b].seq[pick[i]];
            int q[4] = {qq[0], qq[1], qq[2], qq[3]};
            if (rigid_transform(c, base_set[b].ids, q, T.data(), P.data())) {
                c->all_transforms.push_back(T);
                c->all_pose.push_back(P);
                c->all_base_index.push_back((int)b);
            }
        }
        total_quads += nq;
    }
    auto t2 = clk::now();
    float max_score = 0;
    int index = -1;
    for (size_t i = 0; i < c->all_transforms.size(); ++i) {
        float l = lcp_score(c, c->all_transforms[i].data(), NULL, NULL);
        if (l > max_score) { max_score = l; index = (int)i; }
    }
    auto t3 = clk::now();
    out->n_bases = (int)base_set.size();
    out->n_quads_total = total_quads;
    out->n_candidates = (int)c->all_transforms.size();
    out->best_lcp = max_score;
    out->best_index = index;
    for (int i = 0; i < 16; ++i) out->best_pose16[i] = index >= 0 ? c->all_pose[index][i] : 0.0f;
    out->t_sample_s = std::chrono::duration<double>(t1 - t0).count();
    out->t_congruent_s = std::chrono::duration<double>(t2 - t1).count();
    out->t_verify_s = std::chrono::duration<double>(t3 - t2).count();
    return 0;
}
int orc_get_candidates(orc_ctx* c, float* T16, float* pose16, int32_t* base_idx, int cap) {
    int n = (int)c->all_transforms.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (T16) memcpy(T16 + 16 * (size_t)i, c->all_transforms[i].data(), 64);
        if (pose16) memcpy(pose16 + 16 * (size_t)i, c->all_pose[i].data(), 64);
        if (base_idx) base_idx[i] = c->all_base_index[i];
    }
    return n;
}

}  // extern "C"
