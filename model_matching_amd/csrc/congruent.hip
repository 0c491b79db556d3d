// congruent.hip -- congruent 4-point sets on the model for every sampled base (HOT LOOP B).
// Replaces
//   stocs_estimator::find_congruent_sets_on_model         reference src/stocs.cpp:753-869
//   Super4PCS::IndexedNormalSet<Point,3,7,float>          reference include/super4pcs/accelerators/
//                                                         normalset.h:65-151, normalset.hpp:57-214
//   index helpers                                         accelerators/utils.h:139-148
//   PairCreationFunctor::getNormalizedEpsilon             include/super4pcs/pairCreationFunctor.h:141-143
//
// The reference builds, per base, a pointer grid (egSize^3 position cells, each a lazily allocated
// array of 343 std::vectors) over the "intersection" points of the P pairs and queries it once per
// Q pair along a sampled cone of directions.  Here all bases are processed together:
//   1. host: the two PPF keys of the base, the <=128 source buckets of each lookup (CSR ranges),
//      the cone sample table (<= 56 unit vectors from libm's acosf/atanf/sinf/cosf -- per-base
//      scalars, exactly the reference's values);
//   2. gather kernel: P and Q pair lists of all bases, straight out of the device index;
//   3. key kernel: (base, position cell, direction cell) -> 64-bit key per P entry; one
//      rocPRIM radix sort of (key, pair) replaces the pointer grid;
//   4. join kernel, one lane per Q pair: quaternion z->n, rotate the cone samples, de-duplicate the
//      direction cells hit (LDS bitset), binary-search each (position cell, direction cell) key in
//      the base's sorted run, filter on |e_Q - e_P|^2 <= epsilon (sic, Q1), count / emit quads;
//   5. segmented radix sort of the packed quads = the order of the reference's
//      std::set<pair<P index, Q index>> (P and Q lists are in lexicographic (id1,id2) order).
// The join is irregular integer/gather work: HBM/L2-bound, no MFMA.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "stocs_ctx.h"

namespace stocs {

#define STOCS_MAX_CONE 64

struct BaseJob {
    float inv1, inv2, cos_alpha;
    float cell;       // _epsilon of the normal set (unit-cube cell edge)
    int egSize;
    int nb;           // cone samples
    uint32_t p_off, p_len, q_off, q_len;   // runs in the gathered P / Q arrays
    float dirs[STOCS_MAX_CONE][3];
};

struct Segment { uint32_t src, len, dst; };

__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ pairs, const Segment* __restrict__ segs, int nseg,
                                                     uint32_t total, uint32_t* __restrict__ out) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int lo = 0, hi = nseg - 1;  // last segment with dst <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].dst <= e) lo = mid; else hi = mid - 1;
    }
    out[e] = pairs[segs[lo].src + (e - segs[lo].dst)];
}

__device__ __forceinline__ V3 ld3c(const float4* a, int i) { const float4 v = a[i]; return mk3(v.x, v.y, v.z); }

// normalset.h:97-104 + utils.h:139-148: int truncation of coord/epsilon, x fastest
__device__ __forceinline__ int64_t index_pos(V3 p, float cell, int eg) {
    const V3 cp = p / cell;
    return (int64_t)(int)cp.z * eg * eg + ((int64_t)(int)cp.y * eg + (int64_t)(int)cp.x);
}
__device__ __forceinline__ int index_normal(V3 n, float nepsilon) {
    const V3 half = mk3(0.5f, 0.5f, 0.5f);
    const V3 cn = (n / 2.0f + half) / nepsilon;
    return (int)cn.z * 49 + ((int)cn.y * 7 + (int)cn.x);
}

__device__ __forceinline__ int find_base(const uint32_t* __restrict__ off, int nB, uint32_t e) {
    int lo = 0, hi = nB - 1;  // last base with off[b] <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// nset.addElement(p1 + inv1*(p2-p1), (p2-p1).normalized(), i)   stocs.cpp:810-818
__global__ __launch_bounds__(256) void pkey_kernel(const BaseJob* __restrict__ jobs, const uint32_t* __restrict__ p_off, int nB,
                                                   const float4* __restrict__ munit, const uint32_t* __restrict__ P, uint32_t totalP,
                                                   float nepsilon, uint64_t* __restrict__ keys) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= totalP) return;
    const int b = find_base(p_off, nB, e);
    const BaseJob& J = jobs[b];
    const uint32_t pr = P[e];
    const V3 p1 = ld3c(munit, pr >> 16), p2 = ld3c(munit, pr & 0xFFFF);
    const V3 n = normalized3(p2 - p1);
    const V3 pos = p1 + J.inv1 * (p2 - p1);
    const int64_t pc = index_pos(pos, J.cell, J.egSize);
    const int nc = index_normal(n, nepsilon);
    uint64_t key;
    if (nc < 0 || nc >= 343 || pc < 0 || pc >= ((int64_t)1 << 31)) key = ((uint64_t)b << 40) | 0xFFFFFFFFFFull;  // unreachable by queries
    else key = ((uint64_t)b << 40) | ((uint64_t)pc * 343ull + (uint64_t)nc);
    keys[e] = key;
}

// After the sort: direction cell of every P entry, and for every (base, position cell) the run of its
// P entries.  Replaces the pointer grid _grid[pId] -> AngularGrid of normalset.h:87-88.
__global__ __launch_bounds__(256) void cell_ranges_kernel(const uint64_t* __restrict__ keys, uint32_t totalP, const uint32_t* __restrict__ p_off,
                                                          long long NC, uint16_t* __restrict__ pdir, uint32_t* __restrict__ cfirst,
                                                          uint32_t* __restrict__ cend) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= totalP) return;
    const uint64_t key = keys[e];
    const uint64_t kk = key & 0xFFFFFFFFFFull;
    const uint32_t b = (uint32_t)(key >> 40);
    if (kk == 0xFFFFFFFFFFull) { pdir[e] = 0xFFFF; return; }
    const uint64_t pc = kk / 343ull;
    pdir[e] = (uint16_t)(kk - pc * 343ull);
    if (!cfirst) return;
    const uint64_t grp = key - (kk - pc * 343ull);   // (base, position cell) part
    bool first = (e == p_off[b]);
    if (!first) { const uint64_t pk = keys[e - 1]; const uint64_t pkk = pk & 0xFFFFFFFFFFull; first = (pk - (pkk - (pkk / 343ull) * 343ull)) != grp || pkk == 0xFFFFFFFFFFull; }
    bool last = (e + 1 == p_off[b + 1]);
    if (!last) { const uint64_t nk = keys[e + 1]; const uint64_t nkk = nk & 0xFFFFFFFFFFull; last = nkk == 0xFFFFFFFFFFull || (nk - (nkk - (nkk / 343ull) * 343ull)) != grp; }
    if (first) cfirst[(long long)b * NC + (long long)pc] = e;
    if (last) cend[(long long)b * NC + (long long)pc] = e + 1;
}

// Eigen Quaternion::setFromTwoVectors((0,0,1), n) and q * v (see DESIGN.md "numerics")
__device__ __forceinline__ void quat_from_z(V3 n, float q[4]) {
    const V3 v0 = normalized3(mk3(0.f, 0.f, 1.f));
    const V3 v1 = normalized3(n);
    float c = dot3(v1, v0);
    if (c < -1.0f + 1e-5f) {
        // Eigen picks the axis from an SVD here; bit-exact restatement is impossible, axis (1,0,0) is used
        c = c > -1.0f ? c : -1.0f;
        const float w2 = (1.0f + c) * 0.5f;
        const float s = stocs_sqrtf(1.0f - w2);
        q[0] = 1.0f * s; q[1] = 0.0f * s; q[2] = 0.0f * s; q[3] = stocs_sqrtf(w2);
        return;
    }
    const V3 axis = cross3(v0, v1);
    const float s = stocs_sqrtf((1.0f + c) * 2.0f);
    const float invs = 1.0f / s;
    q[0] = axis.x * invs; q[1] = axis.y * invs; q[2] = axis.z * invs; q[3] = s * 0.5f;
}
__device__ __forceinline__ V3 quat_rot(const float q[4], V3 v) {
    const V3 qv = mk3(q[0], q[1], q[2]);
    V3 uv = cross3(qv, v);
    uv = uv + uv;
    return (v + q[3] * uv) + cross3(qv, uv);
}

// one lane per Q pair: stocs.cpp:827-858 + normalset.hpp:166-214
template <bool FILL>
__global__ __launch_bounds__(256) void join_kernel(const BaseJob* __restrict__ jobs, const uint32_t* __restrict__ q_off, int nB,
                                                   const float4* __restrict__ munit, const float4* __restrict__ mpos,
                                                   const uint32_t* __restrict__ Q, uint32_t totalQ, const uint64_t* __restrict__ pkeys,
                                                   const uint32_t* __restrict__ pvals, const uint16_t* __restrict__ pdir,
                                                   const uint32_t* __restrict__ cfirst, const uint32_t* __restrict__ cend, long long NC,
                                                   float nepsilon, float dist_thr,
                                                   unsigned long long* __restrict__ qcnt, const unsigned long long* __restrict__ qoff_e, int id_bits,
                                                   uint64_t* __restrict__ quads) {
    __shared__ uint32_t seen[256][11];  // 343-bit set per lane
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= totalQ) return;
    const int b = find_base(q_off, nB, e);
    const BaseJob& J = jobs[b];
    if (J.p_len == 0 || J.nb == 0) { if (!FILL) qcnt[e] = 0; return; }
    const uint32_t qr = Q[e];
    const int qa = qr >> 16, qb = qr & 0xFFFF;
    const V3 p1 = ld3c(munit, qa), p2 = ld3c(munit, qb);
    const V3 pq1 = ld3c(mpos, qa), pq2 = ld3c(mpos, qb);
    const V3 query = p1 + J.inv2 * (p2 - p1);
    const V3 queryQ = pq1 + J.inv2 * (pq2 - pq1);
    const V3 queryn = normalized3(p2 - p1);
    const int64_t pc = index_pos(query, J.cell, J.egSize);
    if (pc < 0 || pc >= ((int64_t)1 << 31)) { if (!FILL) qcnt[e] = 0; return; }
    // the run of P entries that live in this query's position cell (only that cell is inspected, Q9)
    uint32_t lo, hi;
    if (cfirst) {
        if (pc >= NC) { if (!FILL) qcnt[e] = 0; return; }
        lo = cfirst[(long long)b * NC + pc];
        hi = cend[(long long)b * NC + pc];
    } else {
        const uint64_t* keys = pkeys + J.p_off;
        const uint64_t k0 = ((uint64_t)b << 40) | ((uint64_t)pc * 343ull), k1 = k0 + 343ull;
        uint32_t l = 0, h = J.p_len;
        while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] < k0) l = mid + 1; else h = mid; }
        lo = J.p_off + l;
        h = J.p_len;
        while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] < k1) l = mid + 1; else h = mid; }
        hi = J.p_off + l;
    }
    if (lo >= hi) { if (!FILL) qcnt[e] = 0; return; }
    // direction cells hit by the sampled cone (std::set<unsigned> colored of normalset.hpp:188-204)
    uint32_t* my = seen[threadIdx.x];
#pragma unroll
    for (int k = 0; k < 11; ++k) my[k] = 0;
    float q[4];
    quat_from_z(queryn, q);
    for (int a = 0; a < J.nb; ++a) {
        const V3 dir = normalized3(quat_rot(q, mk3(J.dirs[a][0], J.dirs[a][1], J.dirs[a][2])));
        const int id = index_normal(dir, nepsilon);
        if (id < 0 || id >= 343) continue;  // std::array::at would throw (NaN direction)
        my[id >> 5] |= 1u << (id & 31);
    }
    // one linear pass over the position cell's P entries against the direction bitset
    unsigned long long local = 0;
    const unsigned long long out0 = FILL ? qoff_e[e] : 0ull;   // exclusive scan of the count pass: no atomics in the fill pass
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t dc = pdir[k];
        if (dc >= 343u || !((my[dc >> 5] >> (dc & 31)) & 1u)) continue;
        const uint32_t pr = pvals[k];
        const int pa = pr >> 16, pb = pr & 0xFFFF;
        const V3 pp1 = ld3c(mpos, pa), pp2 = ld3c(mpos, pb);
        const V3 invPoint = pp1 + (pp2 - pp1) * J.inv1;
        if (sqn3(queryQ - invPoint) <= dist_thr) {  // squared metres vs metres (Q1), reproduced
            if (FILL)   // sort key: base, then (P.first, P.second, Q.first, Q.second) == the std::set order
                quads[out0 + local] = ((uint64_t)b << (4 * id_bits)) | ((uint64_t)pa << (3 * id_bits)) | ((uint64_t)pb << (2 * id_bits)) |
                                      ((uint64_t)qa << id_bits) | (uint64_t)qb;
            local++;
        }
    }
    if (!FILL) qcnt[e] = local;
}

// 16-bit fallback layout (|M| > 8192 or very many bases): (a,b,c,d) x 16 bits, sorted per base segment
struct XformJobC { int32_t s[4]; int32_t q[4]; };
// picks (base, rank) -> transform jobs, straight from the device-resident sorted quads
__global__ __launch_bounds__(256) void make_jobs_kernel(const uint64_t* __restrict__ quads, const unsigned long long* __restrict__ quad_off,
                                                        const int2* __restrict__ picks, int n, const int32_t* __restrict__ base_ids,
                                                        int id_bits, XformJobC* __restrict__ jobs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int b = picks[j].x;
    const uint64_t key = quads[quad_off[b] + (unsigned long long)picks[j].y];
    const uint64_t m = (1ull << id_bits) - 1ull;
    XformJobC job;
#pragma unroll
    for (int k = 0; k < 4; ++k) job.s[k] = base_ids[4 * b + k];
    job.q[0] = (int)((key >> (3 * id_bits)) & m); job.q[1] = (int)((key >> (2 * id_bits)) & m);
    job.q[2] = (int)((key >> id_bits) & m); job.q[3] = (int)(key & m);
    jobs[j] = job;
}

template <class T>
struct DevBuf {
    T* p;
    DevBuf() : p(NULL) {}
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (p) { (void)hipFree(p); p = NULL; }
        STOCS_HIP_CHECK(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)));
        return STOCS_OK;
    }
};

}  // namespace stocs

using namespace stocs;

extern "C" {

static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
#define STOCS_TICK(label)                                                                          \
    if (dbg) { (void)hipStreamSynchronize(c->stream); const double t_ = now_s(); fprintf(stderr, "[stocs congruent] %-18s %8.3f ms\n", label, (t_ - tprev) * 1e3); tprev = t_; }

int stocs_find_congruent_all(stocs_ctx* c, int64_t* total_quads) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    const bool dbg = getenv("STOCS_DEBUG_TIMING") != NULL;
    double tprev = now_s();
    if (!c->index.built) { set_error("stocs_find_congruent_all: PPF index not built"); return STOCS_ERR_STATE; }
    const int nB = (int)c->bases.size();
    c->quad_off.assign(nB + 1, 0);
    c->quad_id_bits = 16;
    if (total_quads) *total_quads = 0;
    if (nB == 0) return STOCS_OK;
    if (nB >= (1 << 20)) { set_error("too many bases"); return STOCS_ERR_INVALID; }
    const PpfIndex& ix = c->index;

    // ---- 1. host preparation per base ----
    std::vector<BaseJob> jobs(nB);
    std::vector<Segment> psegs, qsegs;
    std::vector<uint32_t> p_off(nB + 1, 0), q_off(nB + 1, 0);
    const float eps_unit = c->prm.distance_threshold / c->ratio;  // getNormalizedEpsilon, pairCreationFunctor.h:141-143
    const int gridDepth = (int)(-log2f(eps_unit));                // normalset.h:117
    const int egSize = (int)pow(2.0, (double)gridDepth);          // :118
    const float cell = 1.f / egSize;                               // :119
    const float nepsilon = (float)((double)(1.0f / 7.0f) + 0.00001);  // normalset.h:86
    uint64_t totP = 0, totQ = 0;
    for (int b = 0; b < nB; ++b) {
        const BaseRec& B = c->bases[b];
        BaseJob& J = jobs[b];
        memset(&J, 0, sizeof(J));
        J.inv1 = B.inv1; J.inv2 = B.inv2;
        J.cell = cell; J.egSize = egSize;
        int K1[4], K2[4];
        ppf_compute(c->h_spos[B.ids[0]], c->h_snrm[B.ids[0]], c->h_spos[B.ids[1]], c->h_snrm[B.ids[1]], ix.tr, ix.rot, K1);  // stocs.cpp:771
        ppf_compute(c->h_spos[B.ids[2]], c->h_snrm[B.ids[2]], c->h_spos[B.ids[3]], c->h_snrm[B.ids[3]], ix.tr, ix.rot, K2);  // stocs.cpp:772
        std::vector<std::pair<uint32_t, uint32_t> > pr, qr;
        plan_lookup(ix, K1, &pr);
        plan_lookup(ix, K2, &qr);
        uint64_t np = 0, nq = 0;
        for (size_t r = 0; r < pr.size(); ++r) np += pr[r].second - pr[r].first;
        for (size_t r = 0; r < qr.size(); ++r) nq += qr[r].second - qr[r].first;
        if (np == 0 || nq == 0) { np = 0; nq = 0; pr.clear(); qr.clear(); }  // stocs.cpp:788
        J.p_off = (uint32_t)totP; J.p_len = (uint32_t)np; J.q_off = (uint32_t)totQ; J.q_len = (uint32_t)nq;
        uint32_t d = (uint32_t)totP;
        for (size_t r = 0; r < pr.size(); ++r) { Segment s = {pr[r].first, pr[r].second - pr[r].first, d}; psegs.push_back(s); d += s.len; }
        d = (uint32_t)totQ;
        for (size_t r = 0; r < qr.size(); ++r) { Segment s = {qr[r].first, qr[r].second - qr[r].first, d}; qsegs.push_back(s); d += s.len; }
        p_off[b] = (uint32_t)totP; q_off[b] = (uint32_t)totQ;
        totP += np; totQ += nq;
        if (totP >= 0xFFFF0000ull || totQ >= 0xFFFF0000ull) { set_error("pair lists exceed 2^32 entries"); return STOCS_ERR_CAPACITY; }
        // cone table: normalset.hpp:178-190 (float libm calls, per base)
        J.cos_alpha = dot3(normalized3(c->h_spos[B.ids[1]] - c->h_spos[B.ids[0]]), normalized3(c->h_spos[B.ids[3]] - c->h_spos[B.ids[2]]));  // stocs.cpp:801-803
        const float alpha = acosf(J.cos_alpha);
        const float perimeter = (float)((double)2.0f * M_PI * (double)atanf(alpha));  // sic (Q10)
        const unsigned nb = (unsigned)(2 * ceilf(perimeter * 7.0f / 2.0f));
        const float angleStep = (float)((double)2.0f * M_PI / (double)(float)nb);
        const float sinAlpha = sinf(alpha);
        J.nb = (nb > STOCS_MAX_CONE || !(alpha == alpha)) ? 0 : (int)nb;  // nb <= 56 for any alpha in [0, pi]; NaN alpha -> no samples
        for (int a = 0; a < J.nb; ++a) {
            const float theta = (float)a * angleStep;
            J.dirs[a][0] = sinAlpha * cosf(theta);
            J.dirs[a][1] = sinAlpha * sinf(theta);
            J.dirs[a][2] = J.cos_alpha;
        }
    }
    p_off[nB] = (uint32_t)totP; q_off[nB] = (uint32_t)totQ;
    STOCS_TICK("host prep")
    if (dbg) fprintf(stderr, "[stocs congruent] totP %llu totQ %llu segs %zu %zu\n", (unsigned long long)totP, (unsigned long long)totQ, psegs.size(), qsegs.size());
    if (totP == 0 || totQ == 0) return STOCS_OK;

    // ---- 2-3. gather + keys + sort ----
    DevBuf<BaseJob> d_jobs; DevBuf<Segment> d_psegs, d_qsegs; DevBuf<uint32_t> d_poff, d_qoff, d_P, d_Q, d_Ps;
    DevBuf<uint64_t> d_keys, d_keys_s; DevBuf<char> d_tmp;
    int rc;
    if ((rc = d_jobs.alloc(nB)) || (rc = d_psegs.alloc(psegs.size())) || (rc = d_qsegs.alloc(qsegs.size())) || (rc = d_poff.alloc(nB + 1)) ||
        (rc = d_qoff.alloc(nB + 1)) || (rc = d_P.alloc(totP)) || (rc = d_Q.alloc(totQ)) || (rc = d_Ps.alloc(totP)) || (rc = d_keys.alloc(totP)) ||
        (rc = d_keys_s.alloc(totP)))
        return rc;
    hipStream_t st = c->stream;
    STOCS_HIP_CHECK(hipMemcpyAsync(d_jobs.p, jobs.data(), sizeof(BaseJob) * nB, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_psegs.p, psegs.data(), sizeof(Segment) * psegs.size(), hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_qsegs.p, qsegs.data(), sizeof(Segment) * qsegs.size(), hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_poff.p, p_off.data(), 4 * (nB + 1), hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_qoff.p, q_off.data(), 4 * (nB + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, ix.d_pairs, d_psegs.p, (int)psegs.size(), (uint32_t)totP, d_P.p);
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((totQ + 255) / 256)), dim3(256), 0, st, ix.d_pairs, d_qsegs.p, (int)qsegs.size(), (uint32_t)totQ, d_Q.p);
    hipLaunchKernelGGL(pkey_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, d_jobs.p, d_poff.p, nB, c->d_munit, d_P.p, (uint32_t)totP, nepsilon, d_keys.p);
    STOCS_HIP_CHECK(hipGetLastError());
    size_t tmp_bytes = 0;
    STOCS_HIP_CHECK(rocprim::radix_sort_pairs(NULL, tmp_bytes, d_keys.p, d_keys_s.p, d_P.p, d_Ps.p, (size_t)totP, 0, 64, st));
    if ((rc = d_tmp.alloc(tmp_bytes))) return rc;
    STOCS_HIP_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys.p, d_keys_s.p, d_P.p, d_Ps.p, (size_t)totP, 0, 64, st));

    // direction cells + per-(base, position cell) runs
    const long long NC = (long long)egSize * egSize * egSize;
    const bool use_table = NC > 0 && NC * (long long)nB <= (long long)32 * 1024 * 1024;
    DevBuf<uint16_t> d_pdir; DevBuf<uint32_t> d_cfirst, d_cend;
    if ((rc = d_pdir.alloc(totP))) return rc;
    if (use_table) {
        if ((rc = d_cfirst.alloc((size_t)(NC * nB))) || (rc = d_cend.alloc((size_t)(NC * nB)))) return rc;
        STOCS_HIP_CHECK(hipMemsetAsync(d_cfirst.p, 0, 4 * (size_t)(NC * nB), st));
        STOCS_HIP_CHECK(hipMemsetAsync(d_cend.p, 0, 4 * (size_t)(NC * nB), st));
    }
    hipLaunchKernelGGL(cell_ranges_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, d_keys_s.p, (uint32_t)totP, d_poff.p, NC, d_pdir.p,
                       use_table ? d_cfirst.p : (uint32_t*)NULL, use_table ? d_cend.p : (uint32_t*)NULL);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_TICK("gather+keys+sort")
    // ---- 4. join: count pass, exclusive scan, fill pass (no atomics) ----
    int id_bits = 1;
    while ((1 << id_bits) < c->nM) id_bits++;
    int base_bits = 1;
    while ((1 << base_bits) < nB) base_bits++;
    if (4 * id_bits + base_bits > 64) { set_error("|M| = %d with %d bases does not fit the 64-bit quad key", c->nM, nB); return STOCS_ERR_CAPACITY; }
    DevBuf<unsigned long long> d_qcnt, d_qoffe;   // 64-bit: the total can exceed 2^32 before the capacity check
    if ((rc = d_qcnt.alloc(totQ + 1)) || (rc = d_qoffe.alloc(totQ + 1))) return rc;
    const dim3 jgrid((unsigned)((totQ + 255) / 256));
    hipLaunchKernelGGL(join_kernel<false>, jgrid, dim3(256), 0, st, d_jobs.p, d_qoff.p, nB, c->d_munit, c->d_mpos, d_Q.p, (uint32_t)totQ, d_keys_s.p, d_Ps.p,
                       d_pdir.p, use_table ? d_cfirst.p : (const uint32_t*)NULL, use_table ? d_cend.p : (const uint32_t*)NULL, NC,
                       nepsilon, c->prm.distance_threshold, d_qcnt.p, (const unsigned long long*)NULL, id_bits, (uint64_t*)NULL);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(hipMemsetAsync(d_qcnt.p + totQ, 0, 8, st));
    size_t tmp_scan = 0;
    STOCS_HIP_CHECK(rocprim::exclusive_scan(NULL, tmp_scan, d_qcnt.p, d_qoffe.p, 0ull, (size_t)totQ + 1, rocprim::plus<unsigned long long>(), st));
    DevBuf<char> d_tmp_scan;
    if ((rc = d_tmp_scan.alloc(tmp_scan))) return rc;
    STOCS_HIP_CHECK(rocprim::exclusive_scan(d_tmp_scan.p, tmp_scan, d_qcnt.p, d_qoffe.p, 0ull, (size_t)totQ + 1, rocprim::plus<unsigned long long>(), st));
    // per-base offsets = scan value at the first Q entry of each base
    std::vector<unsigned long long> qoff_at(nB + 1);
    for (int b = 0; b <= nB; ++b)
        STOCS_HIP_CHECK(hipMemcpyAsync(&qoff_at[b], d_qoffe.p + q_off[b], 8, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    STOCS_TICK("join count+scan")
    c->quad_off.assign(nB + 1, 0);
    for (int b = 0; b <= nB; ++b) c->quad_off[b] = qoff_at[b];
    const unsigned long long totQuads = c->quad_off[nB];
    if (totQuads > (1ull << 31)) {  // 16 GiB of packed quads + as much sort scratch
        if (total_quads) *total_quads = (int64_t)totQuads;
        c->quad_off.assign(nB + 1, 0);
        set_error("%llu congruent quads: more than 2^31, refusing to materialise them", totQuads);
        return STOCS_ERR_CAPACITY;
    }
    c->quad_id_bits = id_bits;
    if (total_quads) *total_quads = (int64_t)totQuads;
    if (c->d_quads) { (void)hipFree(c->d_quads); c->d_quads = NULL; }
    if (c->d_quad_off) { (void)hipFree(c->d_quad_off); c->d_quad_off = NULL; }
    STOCS_HIP_CHECK(hipMalloc((void**)&c->d_quad_off, 8 * (size_t)(nB + 1)));
    STOCS_HIP_CHECK(hipMemcpyAsync(c->d_quad_off, c->quad_off.data(), 8 * (size_t)(nB + 1), hipMemcpyHostToDevice, st));
    if (totQuads == 0) { STOCS_HIP_CHECK(hipStreamSynchronize(st)); return STOCS_OK; }
    DevBuf<uint64_t> d_quads;
    if ((rc = d_quads.alloc(totQuads))) return rc;
    STOCS_HIP_CHECK(hipMalloc((void**)&c->d_quads, 8 * (size_t)totQuads));
    hipLaunchKernelGGL(join_kernel<true>, jgrid, dim3(256), 0, st, d_jobs.p, d_qoff.p, nB, c->d_munit, c->d_mpos, d_Q.p, (uint32_t)totQ, d_keys_s.p, d_Ps.p,
                       d_pdir.p, use_table ? d_cfirst.p : (const uint32_t*)NULL, use_table ? d_cend.p : (const uint32_t*)NULL, NC,
                       nepsilon, c->prm.distance_threshold, (unsigned long long*)NULL, d_qoffe.p, id_bits, d_quads.p);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_TICK("join fill")
    // ---- 5. ONE global radix sort on (base, a, b, c, d): per base the order of the reference's std::set ----
    size_t tmp2 = 0;
    const unsigned end_bit = (unsigned)(4 * id_bits + base_bits);
    STOCS_HIP_CHECK(rocprim::radix_sort_keys(NULL, tmp2, d_quads.p, c->d_quads, (size_t)totQuads, 0, end_bit, st));
    DevBuf<char> d_tmp2;
    if ((rc = d_tmp2.alloc(tmp2))) return rc;
    STOCS_HIP_CHECK(rocprim::radix_sort_keys(d_tmp2.p, tmp2, d_quads.p, c->d_quads, (size_t)totQuads, 0, end_bit, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    STOCS_TICK("sort")
    return STOCS_OK;
}

int stocs_get_quads(stocs_ctx* c, int slot, int32_t* quads4, int64_t cap, int64_t* n) {
    if (!c || !n || slot < 0) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (slot + 1 >= (int)c->quad_off.size()) { set_error("stocs_get_quads: no such base slot (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    *n = (int64_t)(c->quad_off[slot + 1] - c->quad_off[slot]);
    if (!quads4 || *n == 0) return STOCS_OK;
    const int64_t m = std::min<int64_t>(*n, cap);
    std::vector<uint64_t> q((size_t)std::max<int64_t>(m, 0));
    if (m > 0) {
        STOCS_HIP_CHECK(hipMemcpyAsync(q.data(), c->d_quads + c->quad_off[slot], 8 * (size_t)m, hipMemcpyDeviceToHost, c->stream));
        STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    const int bits = c->quad_id_bits;
    const uint64_t mask = (1ull << bits) - 1ull;
    for (int64_t i = 0; i < m; ++i) {
        quads4[4 * i + 0] = (int32_t)((q[i] >> (3 * bits)) & mask);
        quads4[4 * i + 1] = (int32_t)((q[i] >> (2 * bits)) & mask);
        quads4[4 * i + 2] = (int32_t)((q[i] >> bits) & mask);
        quads4[4 * i + 3] = (int32_t)(q[i] & mask);
    }
    return (*n > cap) ? STOCS_ERR_CAPACITY : STOCS_OK;
}

// device side of stocs_make_transforms: (base, rank) picks -> XformJob records on the device
int stocs_internal_make_jobs(stocs_ctx* c, const int32_t* picks2_host, int n, void* d_jobs_out) {
    if (n <= 0) return STOCS_OK;
    const int nB = (int)c->bases.size();
    DevBuf<int2> d_picks; DevBuf<int32_t> d_bids;
    int rc;
    if ((rc = d_picks.alloc(n)) || (rc = d_bids.alloc((size_t)nB * 4))) return rc;
    std::vector<int32_t> bids((size_t)nB * 4);
    for (int b = 0; b < nB; ++b) for (int k = 0; k < 4; ++k) bids[4 * b + k] = c->bases[b].ids[k];
    STOCS_HIP_CHECK(hipMemcpyAsync(d_picks.p, picks2_host, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_bids.p, bids.data(), 16 * (size_t)nB, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(make_jobs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_quads, c->d_quad_off, d_picks.p, n, d_bids.p,
                       c->quad_id_bits, (XformJobC*)d_jobs_out);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

}  // extern "C"
