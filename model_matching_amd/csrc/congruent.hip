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
//   4. join, one lane per Q pair: quaternion z->n, rotate the cone samples, de-duplicate the direction
//      cells hit (LDS bitset), walk the P run of the query's position cell, filter on
//      |e_Q - e_P|^2 <= epsilon (sic, Q1).  Only the COUNT pass runs here (+ an exclusive scan);
//   5. quads are produced on demand: a base with fewer than the per-base maximum is materialised and
//      radix-sorted into the order of the reference's std::set<pair<P index, Q index>>; a base with
//      more is only ever sampled, and each sampled rank is resolved by re-running the join of the one Q
//      pair that owns it (resolve_picks_kernel) -- the 10^7-10^8 quads of a Cm trial are never written.
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

struct Segment { uint32_t src, len, dst, base; };

// Pair lists of all bases out of the device index, as (base << 32 | pair) keys.  A lookup is the union of <= 128
// buckets (ppf_index.hip); the reference's bucket vector is in insertion = lexicographic (id1, id2) order
// (rgbd.cpp:123-154), which one radix sort of these keys restores (pair = id1 << 16 | id2).
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ pairs, const Segment* __restrict__ segs, int nseg,
                                                     uint32_t total, uint64_t* __restrict__ out) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int lo = 0, hi = nseg - 1;  // last segment with dst <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].dst <= e) lo = mid; else hi = mid - 1;
    }
    out[e] = ((uint64_t)segs[lo].base << 32) | (uint64_t)pairs[segs[lo].src + (e - segs[lo].dst)];
}

// Zero fill as an ordinary kernel on the context's stream (stream-ordered buffers are only ever touched by kernels
// and explicit copies of that stream; see DevBuf).
__global__ __launch_bounds__(256) void zero_u32_kernel(uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ b) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0; if (b) b[i] = 0; }
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
                                                   const float4* __restrict__ munit, const uint64_t* __restrict__ P, uint32_t totalP,
                                                   float nepsilon, uint64_t* __restrict__ keys, uint32_t* __restrict__ P32) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= totalP) return;
    const int b = find_base(p_off, nB, e);
    const BaseJob& J = jobs[b];
    const uint32_t pr = (uint32_t)P[e];
    P32[e] = pr;
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

// One-sort form of the two steps above when (base, cell, pair) fits 64 bits: the merged key orders a base's P
// entries by cell and, inside a cell, by the list order (lexicographic pair) -- what the list-order sort followed
// by the stable cell sort produces, in 8 instead of 11 radix passes and without a value array.
__global__ __launch_bounds__(256) void pkey_merged_kernel(const BaseJob* __restrict__ jobs, const float4* __restrict__ munit,
                                                          const uint64_t* __restrict__ P, uint32_t totalP, float nepsilon, int cell_bits, int id_bits,
                                                          uint64_t* __restrict__ keys) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= totalP) return;
    const uint64_t g = P[e];
    const uint32_t b = (uint32_t)(g >> 32), pr = (uint32_t)g;
    const BaseJob& J = jobs[b];
    const uint32_t ia = pr >> 16, ib = pr & 0xFFFF;
    const V3 p1 = ld3c(munit, ia), p2 = ld3c(munit, ib);
    const V3 n = normalized3(p2 - p1);
    const V3 pos = p1 + J.inv1 * (p2 - p1);
    const int64_t pc = index_pos(pos, J.cell, J.egSize);
    const int nc = index_normal(n, nepsilon);
    const uint64_t cmask = ((uint64_t)1 << cell_bits) - 1ull;
    uint64_t cell = cmask;   // unreachable by queries
    if (!(nc < 0 || nc >= 343 || pc < 0 || pc >= ((int64_t)1 << 31))) { cell = (uint64_t)pc * 343ull + (uint64_t)nc; if (cell >= cmask) cell = cmask; }
    keys[e] = ((uint64_t)b << (cell_bits + 2 * id_bits)) | (cell << (2 * id_bits)) | ((uint64_t)ia << id_bits) | (uint64_t)ib;
}
__global__ __launch_bounds__(256) void punpack_kernel(const uint64_t* __restrict__ mk, uint32_t totalP, int cell_bits, int id_bits,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ P32) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= totalP) return;
    const uint64_t k = mk[e];
    const uint64_t cmask = ((uint64_t)1 << cell_bits) - 1ull, imask = ((uint64_t)1 << id_bits) - 1ull;
    const uint64_t b = k >> (cell_bits + 2 * id_bits), cell = (k >> (2 * id_bits)) & cmask;
    keys[e] = (b << 40) | (cell == cmask ? 0xFFFFFFFFFFull : cell);
    P32[e] = (uint32_t)((((k >> id_bits) & imask) << 16) | (k & imask));
}

// After the sort: direction cell of every P entry, and for every (base, position cell) the run of its
// P entries.  Replaces the pointer grid _grid[pId] -> AngularGrid of normalset.h:87-88.
__global__ __launch_bounds__(256) void cell_ranges_kernel(const uint64_t* __restrict__ keys, uint32_t totalP, const uint32_t* __restrict__ p_off,
                                                          long long NC, uint16_t* __restrict__ pdir, uint32_t* __restrict__ cfirst,
                                                          uint32_t* __restrict__ cend, const BaseJob* __restrict__ jobs,
                                                          const uint32_t* __restrict__ pvals, const float4* __restrict__ mpos,
                                                          float4* __restrict__ pinv) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= totalP) return;
    const uint64_t key = keys[e];
    const uint64_t kk = key & 0xFFFFFFFFFFull;
    const uint32_t b = (uint32_t)(key >> 40);
    {   // invPoint of stocs.cpp:845-849, once per P entry instead of once per (Q, P) test
        const uint32_t pr = pvals[e];
        const V3 pp1 = ld3c(mpos, pr >> 16), pp2 = ld3c(mpos, pr & 0xFFFF);
        const V3 ip = pp1 + (pp2 - pp1) * jobs[b].inv1;
        pinv[e] = make_float4(ip.x, ip.y, ip.z, 0.f);
    }
    if (kk == 0xFFFFFFFFFFull) { pdir[e] = 0xFFFF; return; }
    const uint64_t pc = kk / 343ull;
    pdir[e] = (uint16_t)(kk - pc * 343ull);
    if (!cfirst) return;
    const uint64_t grp = key - (kk - pc * 343ull);   // (base, position cell) part
    bool first = (e == p_off[b]);
    if (!first) { const uint64_t pk = keys[e - 1]; const uint64_t pkk = pk & 0xFFFFFFFFFFull; first = (pk - (pkk - (pkk / 343ull) * 343ull)) != grp || pkk == 0xFFFFFFFFFFull; }
    bool last = (e + 1 == p_off[b + 1]);
    if (!last) { const uint64_t nk = keys[e + 1]; const uint64_t nkk = nk & 0xFFFFFFFFFFull; last = nkk == 0xFFFFFFFFFFull || (nk - (nkk - (nkk / 343ull) * 343ull)) != grp; }
    if ((long long)pc >= NC) return;   // outside the table (cannot happen for points of the unit cube): never queried
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

// Everything the join needs, by value.
struct JoinArgs {
    const BaseJob* jobs; const uint32_t* q_off; int nB;
    const float4* munit; const float4* mpos;
    const uint64_t* Q; uint32_t totQ;   // (base << 32 | pair), each base's run in lexicographic pair order
    const uint64_t* pkeys; const uint32_t* pvals; const uint16_t* pdir; const float4* pinv;
    const uint32_t* cfirst; const uint32_t* cend; long long NC;
    float nepsilon, dist_thr;
    int id_bits;
};

// The join of ONE Q pair against the P entries of its position cell: stocs.cpp:827-858 + normalset.hpp:166-214.
// Matches come out in the order the reference inserts them into `comb`: direction cells ascending
// (std::set<unsigned> colored), then the P entries of that cell in insertion (= index) order -- the sorted
// (position cell, direction cell, P index) run delivers exactly that.
//   MODE 0: count;  MODE 1: write every match to out[0..];  MODE 2: write the `want`-th match to out[0].
template <int MODE>
__device__ __forceinline__ unsigned long long join_one(const JoinArgs& A, uint32_t e, int b, uint32_t* my, unsigned long long want,
                                                       uint64_t* __restrict__ out) {
    const BaseJob& J = A.jobs[b];
    if (J.p_len == 0 || J.nb == 0) return 0;
    const uint32_t qr = (uint32_t)A.Q[e];
    const int qa = qr >> 16, qb = qr & 0xFFFF;
    const V3 p1 = ld3c(A.munit, qa), p2 = ld3c(A.munit, qb);
    const V3 pq1 = ld3c(A.mpos, qa), pq2 = ld3c(A.mpos, qb);
    const V3 query = p1 + J.inv2 * (p2 - p1);
    const V3 queryQ = pq1 + J.inv2 * (pq2 - pq1);
    const V3 queryn = normalized3(p2 - p1);
    const int64_t pc = index_pos(query, J.cell, J.egSize);
    if (pc < 0 || pc >= ((int64_t)1 << 31)) return 0;
    // the run of P entries that live in this query's position cell (only that cell is inspected, Q9)
    uint32_t lo, hi;
    if (A.cfirst) {
        if (pc >= A.NC) return 0;
        lo = A.cfirst[(long long)b * A.NC + pc];
        hi = A.cend[(long long)b * A.NC + pc];
    } else {
        const uint64_t* keys = A.pkeys + J.p_off;
        const uint64_t k0 = ((uint64_t)b << 40) | ((uint64_t)pc * 343ull), k1 = k0 + 343ull;
        uint32_t l = 0, h = J.p_len;
        while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] < k0) l = mid + 1; else h = mid; }
        lo = J.p_off + l;
        h = J.p_len;
        while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] < k1) l = mid + 1; else h = mid; }
        hi = J.p_off + l;
    }
    if (lo >= hi) return 0;
    // direction cells hit by the sampled cone (std::set<unsigned> colored of normalset.hpp:188-204)
#pragma unroll
    for (int k = 0; k < 11; ++k) my[k] = 0;
    float q[4];
    quat_from_z(queryn, q);
    for (int a = 0; a < J.nb; ++a) {
        const V3 dir = normalized3(quat_rot(q, mk3(J.dirs[a][0], J.dirs[a][1], J.dirs[a][2])));
        const int id = index_normal(dir, A.nepsilon);
        if (id < 0 || id >= 343) continue;  // std::array::at would throw (NaN direction)
        my[id >> 5] |= 1u << (id & 31);
    }
    // one linear pass over the position cell's P entries against the direction bitset
    unsigned long long local = 0;
    const int id_bits = A.id_bits;
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t dc = A.pdir[k];
        if (dc >= 343u || !((my[dc >> 5] >> (dc & 31)) & 1u)) continue;
        const V3 invPoint = ld3c(A.pinv, k);
        if (sqn3(queryQ - invPoint) <= A.dist_thr) {  // squared metres vs metres (Q1), reproduced
            if (MODE != 0) {   // sort key: base, then (P.first, P.second, Q.first, Q.second) == the std::set order
                const uint32_t pr = A.pvals[k];
                const int pa = pr >> 16, pb = pr & 0xFFFF;
                const uint64_t key = ((uint64_t)b << (4 * id_bits)) | ((uint64_t)pa << (3 * id_bits)) | ((uint64_t)pb << (2 * id_bits)) |
                                     ((uint64_t)qa << id_bits) | (uint64_t)qb;
                if (MODE == 1) out[local] = key;
                if (MODE == 2 && local == want) { out[0] = key; return local + 1; }
            }
            local++;
        }
    }
    return local;
}

// (base, position cell) of every Q pair: the count pass walks the Q pairs in this order, so that the lanes of a
// wavefront share one P run (same loop length, broadcast loads) instead of 64 unrelated ones
__global__ __launch_bounds__(256) void qcell_kernel(JoinArgs A, int pc_bits, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.totQ) return;
    const int b = find_base(A.q_off, A.nB, e);
    const BaseJob& J = A.jobs[b];
    const uint32_t qr = (uint32_t)A.Q[e];
    const V3 p1 = ld3c(A.munit, qr >> 16), p2 = ld3c(A.munit, qr & 0xFFFF);
    const int64_t pc = index_pos(p1 + J.inv2 * (p2 - p1), J.cell, J.egSize);
    // compact key (base, cell): only the bits that can be set are sorted; an unusable cell sorts behind the base's real ones
    const uint64_t inval = ((uint64_t)1 << pc_bits) - 1ull;
    keys[e] = ((uint64_t)b << pc_bits) | ((pc < 0 || (uint64_t)pc >= inval) ? inval : (uint64_t)pc);
    vals[e] = e;
}

// per-base totals: the scanned count at the first Q pair of every base (one small copy instead of one per base)
__global__ __launch_bounds__(256) void base_offsets_kernel(const unsigned long long* __restrict__ qoffe, const uint32_t* __restrict__ q_off, int n,
                                                           unsigned long long* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) out[b] = qoffe[q_off[b]];
}

// count pass, one lane per Q pair, in (base, position cell) order; counts land at the pair's list position
__global__ __launch_bounds__(256) void join_count_kernel(JoinArgs A, const uint32_t* __restrict__ qperm, unsigned long long* __restrict__ qcnt) {
    __shared__ uint32_t seen[256][11];  // 343-bit set per lane
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.totQ) return;
    if (i == 0) qcnt[A.totQ] = 0;   // the scan runs over totQ + 1 entries so that its last output is the total
    const uint32_t e = qperm[i];
    const int b = find_base(A.q_off, A.nB, e);
    qcnt[e] = join_one<0>(A, e, b, seen[threadIdx.x], 0ull, NULL);
}

// fill pass for the bases whose out_base is not ~0: destinations from the exclusive scan of the counts (no atomics);
// Q pairs without matches leave at once, so materialising a few small bases costs one sweep over the offsets
__global__ __launch_bounds__(256) void join_fill_kernel(JoinArgs A, const unsigned long long* __restrict__ qoffe,
                                                        const unsigned long long* __restrict__ out_base, uint64_t* __restrict__ quads) {
    __shared__ uint32_t seen[256][11];
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.totQ) return;
    const int b = find_base(A.q_off, A.nB, e);
    const unsigned long long ob = out_base[b];
    if (ob == ~0ull) return;
    const unsigned long long o0 = qoffe[e];
    if (qoffe[e + 1] == o0) return;
    join_one<1>(A, e, b, seen[threadIdx.x], 0ull, quads + ob + (o0 - qoffe[A.q_off[b]]));
}

struct XformJobC { int32_t s[4]; int32_t q[4]; };
struct Pick { int32_t base, rank, dst, sorted; };   // sorted: rank counts in the base's materialised, sorted run

__device__ __forceinline__ void store_job(XformJobC* jobs, int dst, const int32_t* base_ids, int b, uint64_t key, int id_bits) {
    const uint64_t m = (1ull << id_bits) - 1ull;
    XformJobC job;
#pragma unroll
    for (int k = 0; k < 4; ++k) job.s[k] = base_ids[4 * b + k];
    job.q[0] = (int)((key >> (3 * id_bits)) & m); job.q[1] = (int)((key >> (2 * id_bits)) & m);
    job.q[2] = (int)((key >> id_bits) & m); job.q[3] = (int)(key & m);
    jobs[dst] = job;
}

// picks -> transform jobs.  A pick is either a rank in a small base's sorted run (all quads are used, in the
// reference's std::set order) or a rank in a big base's EMISSION order (Q pairs in list order, each one's matches
// in join order): the Q pair is found by binary search in the scanned counts and its join is re-run up to
// the wanted match, so the 10^7-10^8 quads of the big bases are never materialised.  One wavefront per pick:
// the cone samples are spread over the lanes (LDS bitset), the P run is tested 64 entries at a time and the
// wanted match is located with ballot / popcount, in run order.
__global__ __launch_bounds__(256) void resolve_picks_kernel(JoinArgs A, const unsigned long long* __restrict__ qoffe, const Pick* __restrict__ picks, int n,
                                                            const uint64_t* __restrict__ sorted_quads, const unsigned long long* __restrict__ sorted_off,
                                                            const int32_t* __restrict__ base_ids, XformJobC* __restrict__ jobs,
                                                            uint64_t* __restrict__ keys_out, unsigned int* __restrict__ n_unresolved) {
    __shared__ uint32_t seen_all[4][12];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + w;
    if (j >= n) return;   // whole wave leaves; no block-wide barrier below
    uint32_t* seen = seen_all[w];
    const Pick pk = picks[j];
    const int b = pk.base;
    uint64_t key = ~0ull;
    if (pk.sorted) {
        key = sorted_quads[sorted_off[b] + (unsigned long long)pk.rank];
    } else {
        const uint32_t q0 = A.q_off[b], q1 = A.q_off[b + 1];
        const unsigned long long target = qoffe[q0] + (unsigned long long)pk.rank;
        uint32_t elo = q0, ehi = q1 - 1;   // last e in [q0, q1) with qoffe[e] <= target
        while (elo < ehi) {
            const uint32_t mid = elo + ((ehi - elo + 1) >> 1);
            if (qoffe[mid] <= target) elo = mid; else ehi = mid - 1;
        }
        unsigned long long want = target - qoffe[elo];
        // ---- the join of Q pair elo, wave-wide (same arithmetic as join_one) ----
        const BaseJob& J = A.jobs[b];
        const uint32_t qr = (uint32_t)A.Q[elo];
        const int qa = qr >> 16, qb = qr & 0xFFFF;
        const V3 p1 = ld3c(A.munit, qa), p2 = ld3c(A.munit, qb);
        const V3 pq1 = ld3c(A.mpos, qa), pq2 = ld3c(A.mpos, qb);
        const V3 query = p1 + J.inv2 * (p2 - p1);
        const V3 queryQ = pq1 + J.inv2 * (pq2 - pq1);
        const V3 queryn = normalized3(p2 - p1);
        const int64_t pc = index_pos(query, J.cell, J.egSize);
        uint32_t lo = 0, hi = 0;
        if (pc < 0 || pc >= ((int64_t)1 << 31) || (A.cfirst && pc >= A.NC)) {
            // a Q pair without a position cell has no matches and is never selected; stay in bounds regardless
        } else if (A.cfirst) {
            lo = A.cfirst[(long long)b * A.NC + pc];
            hi = A.cend[(long long)b * A.NC + pc];
        } else {
            const uint64_t* keys = A.pkeys + J.p_off;
            const uint64_t k0 = ((uint64_t)b << 40) | ((uint64_t)pc * 343ull), k1 = k0 + 343ull;
            uint32_t l = 0, h = J.p_len;
            while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] < k0) l = mid + 1; else h = mid; }
            lo = J.p_off + l;
            h = J.p_len;
            while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] < k1) l = mid + 1; else h = mid; }
            hi = J.p_off + l;
        }
        if (lane < 12) seen[lane] = 0;
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
        float q[4];
        quat_from_z(queryn, q);
        if (lane < J.nb) {
            const V3 dir = normalized3(quat_rot(q, mk3(J.dirs[lane][0], J.dirs[lane][1], J.dirs[lane][2])));
            const int id = index_normal(dir, A.nepsilon);
            if (id >= 0 && id < 343) atomicOr(&seen[id >> 5], 1u << (id & 31));
        }
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
        const int id_bits = A.id_bits;
        for (uint32_t k0 = lo; k0 < hi; k0 += 64) {
            const uint32_t k = k0 + lane;
            bool hit = false;
            int pa = 0, pb = 0;
            if (k < hi) {
                const uint32_t dc = A.pdir[k];
                if (dc < 343u && ((seen[dc >> 5] >> (dc & 31)) & 1u)) {
                    const uint32_t pr = A.pvals[k];
                    pa = pr >> 16; pb = pr & 0xFFFF;
                    hit = sqn3(queryQ - ld3c(A.pinv, k)) <= A.dist_thr;
                }
            }
            const unsigned long long m = __ballot(hit);
            const unsigned long long cnt = (unsigned long long)__popcll(m);
            if (want < cnt) {
                const unsigned long long below = m & ((1ull << lane) - 1ull);
                if (hit && (unsigned long long)__popcll(below) == want)
                    key = ((uint64_t)b << (4 * id_bits)) | ((uint64_t)pa << (3 * id_bits)) | ((uint64_t)pb << (2 * id_bits)) |
                          ((uint64_t)qa << id_bits) | (uint64_t)qb;
                // hand the winner's key to every lane
                const int src = __ffsll((long long)__ballot(key != ~0ull)) - 1;
                key = ((uint64_t)(uint32_t)__shfl((int)(key >> 32), src, 64) << 32) | (uint64_t)(uint32_t)__shfl((int)(key & 0xFFFFFFFFull), src, 64);
                break;
            }
            want -= cnt;
        }
    }
    if (lane == 0) {
        if (key == ~0ull) {   // cannot happen while counts and join agree; never hand an invalid quad to the next kernel
            atomicAdd(n_unresolved, 1u);
            key = 0;          // quad (0,0,0,0): a degenerate frame, rejected by rigid_transform_kernel
        }
        if (jobs) store_job(jobs, pk.dst, base_ids, b, key, A.id_bits);
        if (keys_out) keys_out[pk.dst] = key;
    }
}

// Device memory of this file comes from two arenas (stocs_ctx.h) owned by the context, reset (not freed) at the start
// of the entry point that owns the arena: after the first trial a call does no hipMalloc / hipFree at all.
// (hipMallocAsync pools were tried first and returned stale data under reuse on this ROCm build -- see DESIGN.md.)
static thread_local Arena* tl_arena = NULL;   // set by every entry point of this file before it allocates

template <class T>
struct DevBuf {   // typed view of arena memory; nothing to release
    T* p;
    DevBuf() : p(NULL) {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    int alloc(size_t n) { return tl_arena->take(n * sizeof(T), (void**)&p); }
};

// What stays on the device after the count pass (stocs_find_congruent_all) so that quads can be produced on
// demand: the gathered Q list, the sorted P run with its cell tables and the scanned per-Q match counts.
struct CongruentState {
    Arena arena_state;   // the buffers below + the temporaries of stocs_find_congruent_all; reset by that call
    Arena arena_tmp;     // temporaries of the calls that produce quads afterwards; reset by each of them
    bool valid = false;  // a count pass has completed and its buffers are intact
    int nB = 0;
    uint32_t totP = 0, totQ = 0;
    long long NC = 0;
    bool use_table = false;
    float nepsilon = 0;
    int id_bits = 16, base_bits = 1;
    std::vector<uint32_t> q_off;
    DevBuf<BaseJob> d_jobs;
    DevBuf<uint32_t> d_qoff, d_Ps, d_cfirst, d_cend;
    DevBuf<uint64_t> d_keys_s, d_Q;
    DevBuf<float4> d_pinv;
    DevBuf<uint16_t> d_pdir;
    DevBuf<unsigned long long> d_qoffe;
    DevBuf<int32_t> d_bids;
    DevBuf<unsigned int> d_err;   // picks resolve_picks_kernel could not resolve (internal consistency check)
    JoinArgs args(const stocs_ctx* c) const {
        JoinArgs A;
        A.jobs = d_jobs.p; A.q_off = d_qoff.p; A.nB = nB; A.munit = c->d_munit; A.mpos = c->d_mpos; A.Q = d_Q.p; A.totQ = totQ;
        A.pkeys = d_keys_s.p; A.pvals = d_Ps.p; A.pdir = d_pdir.p; A.pinv = d_pinv.p;
        A.cfirst = use_table ? d_cfirst.p : NULL; A.cend = use_table ? d_cend.p : NULL; A.NC = NC;
        A.nepsilon = nepsilon; A.dist_thr = c->prm.distance_threshold; A.id_bits = id_bits;
        return A;
    }
};

// Materialises the quads of the bases with sel[b] != 0 into one device buffer, sorted by (base, a, b, c, d) =
// per base the order of the reference's std::set<pair<P index, Q index>>.  off[b] .. off[b+1] is base b's run.
static int materialise(stocs_ctx* c, CongruentState* S, const std::vector<char>& sel, DevBuf<uint64_t>* out, std::vector<unsigned long long>* off) {
    const int nB = S->nB;
    std::vector<unsigned long long> out_base(nB, ~0ull);
    off->assign(nB + 1, 0);
    unsigned long long tot = 0;
    for (int b = 0; b < nB; ++b) {
        (*off)[b] = tot;
        if (sel[b]) { out_base[b] = tot; tot += c->quad_off[b + 1] - c->quad_off[b]; }
    }
    (*off)[nB] = tot;
    if (tot == 0) return STOCS_OK;
    if (tot > (1ull << 31)) { set_error("%llu congruent quads requested at once: more than 2^31, refusing to materialise them", tot); return STOCS_ERR_CAPACITY; }
    hipStream_t st = c->stream;
    DevBuf<unsigned long long> d_ob; DevBuf<uint64_t> d_raw; DevBuf<char> d_tmp;
    int rc;
    if ((rc = d_ob.alloc(nB)) || (rc = d_raw.alloc(tot)) || (rc = out->alloc(tot))) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(d_ob.p, out_base.data(), 8 * (size_t)nB, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(join_fill_kernel, dim3((unsigned)((S->totQ + 255) / 256)), dim3(256), 0, st, S->args(c), S->d_qoffe.p, d_ob.p, d_raw.p);
    STOCS_HIP_CHECK(hipGetLastError());
    size_t tmp = 0;
    const unsigned end_bit = (unsigned)(4 * S->id_bits + S->base_bits);
    STOCS_HIP_CHECK(rocprim::radix_sort_keys(NULL, tmp, d_raw.p, out->p, (size_t)tot, 0, end_bit, st));
    if ((rc = d_tmp.alloc(tmp))) return rc;
    STOCS_HIP_CHECK(rocprim::radix_sort_keys(d_tmp.p, tmp, d_raw.p, out->p, (size_t)tot, 0, end_bit, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));   // the temporaries die with this scope
    return STOCS_OK;
}

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

void stocs_internal_free_congruent(stocs_ctx* c) {   // stocs_ctx_destroy: nothing is in flight any more
    if (c && c->cong) {
        CongruentState* S = (CongruentState*)c->cong;
        S->arena_state.destroy(); S->arena_tmp.destroy();
        delete S;
        c->cong = NULL;
    }
}

void stocs_internal_invalidate_congruent(stocs_ctx* c) {   // the counted state refers to bases of another scene
    if (c && c->cong) ((CongruentState*)c->cong)->valid = false;
}

int stocs_find_congruent_all(stocs_ctx* c, int64_t* total_quads) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    const bool dbg = getenv("STOCS_DEBUG_TIMING") != NULL;
    double tprev = now_s();
    if (!c->index.built) { set_error("stocs_find_congruent_all: PPF index not built"); return STOCS_ERR_STATE; }
    const int nB = (int)c->bases.size();
    if (!c->cong) c->cong = new CongruentState();
    CongruentState* S = (CongruentState*)c->cong;
    S->valid = false;
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));   // the previous trial's buffers are about to be reused
    { int rc0 = S->arena_state.reset(); if (rc0) return rc0; }
    tl_arena = &S->arena_state;
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
        for (size_t r = 0; r < pr.size(); ++r) { Segment s = {pr[r].first, pr[r].second - pr[r].first, d, (uint32_t)b}; psegs.push_back(s); d += s.len; }
        d = (uint32_t)totQ;
        for (size_t r = 0; r < qr.size(); ++r) { Segment s = {qr[r].first, qr[r].second - qr[r].first, d, (uint32_t)b}; qsegs.push_back(s); d += s.len; }
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
    {   // everything this call allocates, estimated up front: one slab, one hipMalloc in a context's lifetime (if sizes stay put)
        const long long NCe = (long long)egSize * egSize * egSize;
        const size_t tables = (NCe > 0 && NCe * (long long)nB <= (long long)32 * 1024 * 1024) ? (size_t)(NCe * nB) * 8 : 0;
        int rc0 = S->arena_state.reserve((size_t)totP * 64 + (size_t)totQ * 64 + tables + ((size_t)48 << 20));
        if (rc0) return rc0;
    }
    S->nB = nB; S->totP = (uint32_t)totP; S->totQ = (uint32_t)totQ; S->nepsilon = nepsilon; S->q_off = q_off;
    DevBuf<Segment> d_psegs, d_qsegs; DevBuf<uint32_t> d_poff, d_P32;
    DevBuf<uint64_t> d_keys, d_Pg, d_Pl, d_Qg; DevBuf<char> d_tmp;
    int rc;
    if ((rc = S->d_jobs.alloc(nB)) || (rc = d_psegs.alloc(psegs.size())) || (rc = d_qsegs.alloc(qsegs.size())) || (rc = d_poff.alloc(nB + 1)) ||
        (rc = S->d_qoff.alloc(nB + 1)) || (rc = d_Pg.alloc(totP)) || (rc = d_Pl.alloc(totP)) || (rc = d_Qg.alloc(totQ)) || (rc = S->d_Q.alloc(totQ)) ||
        (rc = d_P32.alloc(totP)) || (rc = S->d_Ps.alloc(totP)) || (rc = d_keys.alloc(totP)) || (rc = S->d_keys_s.alloc(totP)) ||
        (rc = S->d_bids.alloc((size_t)nB * 4)) || (rc = S->d_err.alloc(1)))
        return rc;
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(256), 0, st, S->d_err.p, (size_t)1, (uint32_t*)NULL);
    std::vector<int32_t> bids((size_t)nB * 4);
    for (int b = 0; b < nB; ++b) for (int k = 0; k < 4; ++k) bids[4 * b + k] = c->bases[b].ids[k];
    STOCS_HIP_CHECK(hipMemcpyAsync(S->d_bids.p, bids.data(), 16 * (size_t)nB, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(S->d_jobs.p, jobs.data(), sizeof(BaseJob) * nB, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_psegs.p, psegs.data(), sizeof(Segment) * psegs.size(), hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_qsegs.p, qsegs.data(), sizeof(Segment) * qsegs.size(), hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_poff.p, p_off.data(), 4 * (nB + 1), hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(S->d_qoff.p, q_off.data(), 4 * (nB + 1), hipMemcpyHostToDevice, st));
    int base_bits = 1;
    while ((1 << base_bits) < nB) base_bits++;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, ix.d_pairs, d_psegs.p, (int)psegs.size(), (uint32_t)totP, d_Pg.p);
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((totQ + 255) / 256)), dim3(256), 0, st, ix.d_pairs, d_qsegs.p, (int)qsegs.size(), (uint32_t)totQ, d_Qg.p);
    STOCS_HIP_CHECK(hipGetLastError());
    // direction cells + per-(base, position cell) runs
    const long long NC = (long long)egSize * egSize * egSize;
    const bool use_table = NC > 0 && NC * (long long)nB <= (long long)32 * 1024 * 1024;
    int id_bits = 1;
    while ((1 << id_bits) < c->nM) id_bits++;
    // each base's P and Q list into the reference's list order (lexicographic pairs); bases stay contiguous
    int cell_bits = 1;   // valid P cells of the table path are < NC * 343 (a P entry outside the table is never looked at)
    {
        const unsigned long long cell_lim = use_table ? (unsigned long long)NC * 343ull : (((unsigned long long)1 << 31) * 343ull);
        while (cell_bits < 41 && ((unsigned long long)1 << cell_bits) < cell_lim + 2ull) cell_bits++;
    }
    const bool merged = base_bits + cell_bits + 2 * id_bits <= 64 && cell_bits <= 40 && !getenv("STOCS_P_TWO_SORTS");   // env: keeps the general path testable
    size_t tmp_bytes = 0, tb2 = 0, tb3 = 0;
    STOCS_HIP_CHECK(rocprim::radix_sort_keys(NULL, tmp_bytes, d_Pg.p, d_Pl.p, (size_t)totP, 0, 64, st));
    STOCS_HIP_CHECK(rocprim::radix_sort_keys(NULL, tb2, d_Qg.p, S->d_Q.p, (size_t)totQ, 0, 32 + base_bits, st));
    STOCS_HIP_CHECK(rocprim::radix_sort_pairs(NULL, tb3, d_keys.p, S->d_keys_s.p, d_P32.p, S->d_Ps.p, (size_t)totP, 0, 40 + base_bits, st));
    tmp_bytes = std::max(tmp_bytes, std::max(tb2, tb3));
    if ((rc = d_tmp.alloc(tmp_bytes))) return rc;
    STOCS_HIP_CHECK(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_Qg.p, S->d_Q.p, (size_t)totQ, 0, 32 + base_bits, st));
    if (merged) {
        hipLaunchKernelGGL(pkey_merged_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, S->d_jobs.p, c->d_munit, d_Pg.p, (uint32_t)totP, nepsilon,
                           cell_bits, id_bits, d_keys.p);
        STOCS_HIP_CHECK(hipGetLastError());
        STOCS_HIP_CHECK(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.p, d_Pl.p, (size_t)totP, 0, (unsigned)(base_bits + cell_bits + 2 * id_bits), st));
        hipLaunchKernelGGL(punpack_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, d_Pl.p, (uint32_t)totP, cell_bits, id_bits, S->d_keys_s.p,
                           S->d_Ps.p);
    } else {
        STOCS_HIP_CHECK(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_Pg.p, d_Pl.p, (size_t)totP, 0, 32 + base_bits, st));
        // (base, position cell, direction cell) keys; the stable sort keeps the list order inside a cell
        hipLaunchKernelGGL(pkey_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, S->d_jobs.p, d_poff.p, nB, c->d_munit, d_Pl.p, (uint32_t)totP,
                           nepsilon, d_keys.p, d_P32.p);
        STOCS_HIP_CHECK(hipGetLastError());
        STOCS_HIP_CHECK(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys.p, S->d_keys_s.p, d_P32.p, S->d_Ps.p, (size_t)totP, 0, 40 + base_bits, st));
    }
    STOCS_HIP_CHECK(hipGetLastError());

    S->NC = NC; S->use_table = use_table;
    if ((rc = S->d_pdir.alloc(totP)) || (rc = S->d_pinv.alloc(totP))) return rc;
    if (use_table) {
        if ((rc = S->d_cfirst.alloc((size_t)(NC * nB))) || (rc = S->d_cend.alloc((size_t)(NC * nB)))) return rc;
        hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)(((size_t)(NC * nB) + 255) / 256)), dim3(256), 0, st, S->d_cfirst.p, (size_t)(NC * nB), S->d_cend.p);
    }
    hipLaunchKernelGGL(cell_ranges_kernel, dim3((unsigned)((totP + 255) / 256)), dim3(256), 0, st, S->d_keys_s.p, (uint32_t)totP, d_poff.p, NC, S->d_pdir.p,
                       use_table ? S->d_cfirst.p : (uint32_t*)NULL, use_table ? S->d_cend.p : (uint32_t*)NULL, S->d_jobs.p, S->d_Ps.p, c->d_mpos, S->d_pinv.p);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_TICK("gather+keys+sort")
    // ---- 4. join: count pass + exclusive scan.  The quads themselves are produced on demand (materialise / resolve_picks_kernel) ----
    if (4 * id_bits + base_bits > 64) { set_error("|M| = %d with %d bases does not fit the 64-bit quad key", c->nM, nB); return STOCS_ERR_CAPACITY; }
    S->id_bits = id_bits; S->base_bits = base_bits;
    DevBuf<unsigned long long> d_qcnt;   // 64-bit: the total can exceed 2^32
    DevBuf<uint32_t> d_qperm;
    if ((rc = d_qcnt.alloc(totQ + 1)) || (rc = S->d_qoffe.alloc(totQ + 1))) return rc;
    {   // walk order of the count pass: Q pairs by (base, position cell)
        DevBuf<uint64_t> d_qk, d_qk_s; DevBuf<uint32_t> d_qv;
        if ((rc = d_qk.alloc(totQ)) || (rc = d_qk_s.alloc(totQ)) || (rc = d_qv.alloc(totQ)) || (rc = d_qperm.alloc(totQ))) return rc;
        int pc_bits = 1;   // cells of the table path are < NC; without the table any cell below 2^31 may occur
        const long long pc_lim = use_table ? NC + 1 : ((long long)1 << 31) + 1;
        while (((long long)1 << pc_bits) < pc_lim + 1) pc_bits++;
        hipLaunchKernelGGL(qcell_kernel, dim3((unsigned)((totQ + 255) / 256)), dim3(256), 0, st, S->args(c), pc_bits, d_qk.p, d_qv.p);
        STOCS_HIP_CHECK(hipGetLastError());
        size_t tq = 0;
        STOCS_HIP_CHECK(rocprim::radix_sort_pairs(NULL, tq, d_qk.p, d_qk_s.p, d_qv.p, d_qperm.p, (size_t)totQ, 0, pc_bits + base_bits, st));
        DevBuf<char> d_tq;
        if ((rc = d_tq.alloc(tq))) return rc;
        STOCS_HIP_CHECK(rocprim::radix_sort_pairs(d_tq.p, tq, d_qk.p, d_qk_s.p, d_qv.p, d_qperm.p, (size_t)totQ, 0, pc_bits + base_bits, st));
    }
    hipLaunchKernelGGL(join_count_kernel, dim3((unsigned)((totQ + 255) / 256)), dim3(256), 0, st, S->args(c), d_qperm.p, d_qcnt.p);
    STOCS_HIP_CHECK(hipGetLastError());
    size_t tmp_scan = 0;
    STOCS_HIP_CHECK(rocprim::exclusive_scan(NULL, tmp_scan, d_qcnt.p, S->d_qoffe.p, 0ull, (size_t)totQ + 1, rocprim::plus<unsigned long long>(), st));
    DevBuf<char> d_tmp_scan;
    if ((rc = d_tmp_scan.alloc(tmp_scan))) return rc;
    STOCS_HIP_CHECK(rocprim::exclusive_scan(d_tmp_scan.p, tmp_scan, d_qcnt.p, S->d_qoffe.p, 0ull, (size_t)totQ + 1, rocprim::plus<unsigned long long>(), st));
    // per-base offsets = scan value at the first Q entry of each base
    std::vector<unsigned long long> qoff_at(nB + 1);
    DevBuf<unsigned long long> d_boff;
    if ((rc = d_boff.alloc(nB + 1))) return rc;
    hipLaunchKernelGGL(base_offsets_kernel, dim3((unsigned)((nB + 1 + 255) / 256)), dim3(256), 0, st, S->d_qoffe.p, S->d_qoff.p, nB + 1, d_boff.p);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(hipMemcpyAsync(qoff_at.data(), d_boff.p, 8 * (size_t)(nB + 1), hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    STOCS_TICK("join count+scan")
    for (int b = 0; b <= nB; ++b) c->quad_off[b] = qoff_at[b];
    S->valid = true;
    c->quad_id_bits = id_bits;
    if (total_quads) *total_quads = (int64_t)c->quad_off[nB];
    return STOCS_OK;
}

int stocs_get_quads(stocs_ctx* c, int slot, int32_t* quads4, int64_t cap, int64_t* n) {
    if (!c || !n || slot < 0) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (slot + 1 >= (int)c->quad_off.size()) { set_error("stocs_get_quads: no such base slot (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    *n = (int64_t)(c->quad_off[slot + 1] - c->quad_off[slot]);
    if (!quads4 || *n == 0) return STOCS_OK;
    CongruentState* S = (CongruentState*)c->cong;
    if (!S || !S->valid) { set_error("stocs_get_quads: no congruent state (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    { int rc0 = S->arena_tmp.reset(); if (rc0) return rc0; }
    tl_arena = &S->arena_tmp;
    const int64_t m = std::min<int64_t>(*n, cap);
    std::vector<uint64_t> q((size_t)std::max<int64_t>(m, 0));
    if (m > 0) {
        std::vector<char> sel(S->nB, 0);
        sel[slot] = 1;
        DevBuf<uint64_t> d_sorted;
        std::vector<unsigned long long> off;
        int rc = materialise(c, S, sel, &d_sorted, &off);
        if (rc) return rc;
        STOCS_HIP_CHECK(hipMemcpyAsync(q.data(), d_sorted.p + off[slot], 8 * (size_t)m, hipMemcpyDeviceToHost, c->stream));
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

// quads of base `slot` at the given ranks of its EMISSION order (the order in which the reference's loop inserts
// them into `comb`, stocs.cpp:827-858) -- what stocs_make_transforms samples from when a base has >= max quads
int stocs_get_quads_at(stocs_ctx* c, int slot, const int64_t* ranks, int n, int32_t* quads4) {
    if (!c || slot < 0 || n < 0 || (n && (!ranks || !quads4))) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (slot + 1 >= (int)c->quad_off.size()) { set_error("stocs_get_quads_at: no such base slot (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    if (n == 0) return STOCS_OK;
    CongruentState* S = (CongruentState*)c->cong;
    const int64_t nq = (int64_t)(c->quad_off[slot + 1] - c->quad_off[slot]);
    std::vector<Pick> picks(n);
    for (int i = 0; i < n; ++i) {
        if (ranks[i] < 0 || ranks[i] >= nq || ranks[i] > 0x7FFFFFFFll) { set_error("stocs_get_quads_at: rank %lld out of range (%lld quads)", (long long)ranks[i], (long long)nq); return STOCS_ERR_INVALID; }
        picks[i].base = slot; picks[i].rank = (int32_t)ranks[i]; picks[i].dst = i; picks[i].sorted = 0;
    }
    if (!S || !S->valid) { set_error("stocs_get_quads_at: no congruent state (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    { int rc0 = S->arena_tmp.reset(); if (rc0) return rc0; }
    tl_arena = &S->arena_tmp;
    DevBuf<Pick> d_picks; DevBuf<uint64_t> d_keys;
    int rc;
    if ((rc = d_picks.alloc(n)) || (rc = d_keys.alloc(n))) return rc;
    std::vector<uint64_t> keys(n);
    STOCS_HIP_CHECK(hipMemcpyAsync(d_picks.p, picks.data(), sizeof(Pick) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(resolve_picks_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, S->args(c), S->d_qoffe.p, d_picks.p, n,
                       (const uint64_t*)NULL, (const unsigned long long*)NULL, S->d_bids.p, (XformJobC*)NULL, d_keys.p, S->d_err.p);
    STOCS_HIP_CHECK(hipGetLastError());
    unsigned int n_err = 0;
    STOCS_HIP_CHECK(hipMemcpyAsync(keys.data(), d_keys.p, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(&n_err, S->d_err.p, 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (n_err) { set_error("stocs_get_quads_at: %u ranks could not be resolved (internal inconsistency)", n_err); return STOCS_ERR_STATE; }
    const int bits = c->quad_id_bits;
    const uint64_t mask = (1ull << bits) - 1ull;
    for (int i = 0; i < n; ++i) {
        quads4[4 * i + 0] = (int32_t)((keys[i] >> (3 * bits)) & mask);
        quads4[4 * i + 1] = (int32_t)((keys[i] >> (2 * bits)) & mask);
        quads4[4 * i + 2] = (int32_t)((keys[i] >> bits) & mask);
        quads4[4 * i + 3] = (int32_t)(keys[i] & mask);
    }
    return STOCS_OK;
}

// device side of stocs_make_transforms: picks4 = (base, rank, destination job, sorted?) records -> XformJob records
// on the device.  Bases picked with sorted != 0 are materialised and sorted first (they are the small ones).
int stocs_internal_make_jobs(stocs_ctx* c, const int32_t* picks4_host, int n, void* d_jobs_out) {
    if (n <= 0) return STOCS_OK;
    CongruentState* S = (CongruentState*)c->cong;
    if (!S || !S->valid) { set_error("stocs_make_transforms: no congruent state (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    { int rc0 = S->arena_tmp.reset(); if (rc0) return rc0; }
    tl_arena = &S->arena_tmp;
    const Pick* picks = (const Pick*)picks4_host;
    std::vector<char> sel(S->nB, 0);
    bool any_sorted = false;
    for (int i = 0; i < n; ++i) if (picks[i].sorted) { sel[picks[i].base] = 1; any_sorted = true; }
    DevBuf<uint64_t> d_sorted; DevBuf<unsigned long long> d_soff; DevBuf<Pick> d_picks;
    std::vector<unsigned long long> off;
    int rc;
    if (any_sorted) {
        if ((rc = materialise(c, S, sel, &d_sorted, &off)) || (rc = d_soff.alloc(off.size()))) return rc;
        STOCS_HIP_CHECK(hipMemcpyAsync(d_soff.p, off.data(), 8 * off.size(), hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = d_picks.alloc(n))) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(d_picks.p, picks, sizeof(Pick) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(resolve_picks_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, S->args(c), S->d_qoffe.p, d_picks.p, n, d_sorted.p,
                       d_soff.p, S->d_bids.p, (XformJobC*)d_jobs_out, (uint64_t*)NULL, S->d_err.p);
    STOCS_HIP_CHECK(hipGetLastError());
    unsigned int n_err = 0;
    STOCS_HIP_CHECK(hipMemcpyAsync(&n_err, S->d_err.p, 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (n_err) { set_error("stocs_make_transforms: %u picks could not be resolved (internal inconsistency)", n_err); return STOCS_ERR_STATE; }
    return STOCS_OK;
}

}  // extern "C"
