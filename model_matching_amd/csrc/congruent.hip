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
//   1. host: the two PPF keys of the base, the source buckets of each lookup (CSR ranges of the index, in ascending
//      index position), the cone sample table (<= 56 unit vectors from libm's acosf/atanf/sinf/cosf -- per-base
//      scalars, exactly the reference's values);
//   2. gather_key_kernel: the P and Q pair lists of all bases straight out of the device index, each entry with its
//      32-bit key (base, position cell of its intersection point);
//   3. ONE stable rocPRIM radix sort per list replaces the pointer grid: the Q entries by (base, position cell) in three
//      passes; the P entries by position cell alone in two -- the gather is base-major, so the entries of a (base, cell)
//      stay together, in index order, and the run table finds them;
//   4. p_records_kernel: per P entry a 16-byte record (world-space intersection point, direction cell), the direction
//      cell alone as 2 bytes, and the per-(base, cell) run table;
//   5. join, one lane per Q pair in (base, cell) order: cone -> direction-cell bitset in LDS (filtered exact arithmetic,
//      cone_cells.h), one pass over the P run of the query's cell, |e_Q - e_P|^2 <= epsilon (sic, Q1) -- a gate that cannot
//      fail inside one position cell while 12 epsilon^2 < epsilon, in which case the count reads the direction cells
//      alone.  Only the COUNT pass runs here (+ an exclusive scan);
//   6. quads are produced on demand: a base with fewer than the per-base maximum is materialised and
//      radix-sorted into the order of the reference's std::set<pair<P index, Q index>>; a base with
//      more is only ever sampled, and each sampled rank is resolved by re-running the join of the one Q
//      pair that owns it (resolve_picks_kernel) -- the 10^7-10^8 quads of a Cm trial are never written.
// Enumeration of a base's quads for that sampling ("walk order", a documented divergence like the seeded draw
// itself): by (position cell of the Q pair, index position of the Q pair, index position of the
// P pair), where the index position of a model pair is its place in the PPF index (ascending quantised feature,
// then ascending (id1, id2)).  The oracle enumerates the same way.
// The join is irregular integer/gather work: HBM/L2-bound, no MFMA.
#include <functional>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <cstring>


#include <algorithm>

#include "cone_cells.h"
#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

struct BaseJob {
    float inv1, inv2, cos_alpha;
    float cell;       // _epsilon of the normal set (unit-cube cell edge)
    int egSize;
    int nb;           // cone samples
    uint32_t p_off, p_len, q_off, q_len;   // runs in the gathered P / Q arrays
    float sin_alpha;  // the cone samples of the base are (sin_alpha * cos theta_a, sin_alpha * sin theta_a, cos_alpha), a < nb, with cos / sin of
                      // theta_a = a * angleStep(nb) from ONE table shared by all bases (cone_trig: they depend on nb alone).  Round 3 carried the
                      // 56 products per base in the job record: 808 bytes per base to fill and upload -- 0.7 ms of host time for the 6 000
                      // bases of a 64-trial batch; the one float product per sample is the same IEEE operation on the device.
    float pad;
};

struct Segment { uint32_t src, len, dst, base; };

__device__ __forceinline__ V3 ld3c(const float4* a, int i) { const float4 v = a[i]; return mk3(v.x, v.y, v.z); }

// ---- planning of the lookups on the device (what plan_lookup of ppf_index.hip does per key on the host) ----
// Totals and array sizes of a trial's pair lists, read back by the host (the only thing it needs before it can size the sorts)
struct PlanOut { unsigned long long totP, totQ; uint32_t n_pseg, n_qseg, overflow, pad; };

// One 64-lane workgroup per (base, list): the PPF key of the base's point pair (stocs.cpp:771-772, the reference's double
// arithmetic), its 128 probes of the bucket table in ASCENDING key order -- F = K - o with o0 in {-tr, 0}, ok in
// {-2 rot, -rot, 0, rot}, rgbd.cpp:130-137 read from the query side -- two per lane, then lane 0 merges adjacent buckets
// into ranges exactly as plan_lookup does: the gathered list is in index order.
__global__ __launch_bounds__(64) void plan_ranges_kernel(const uint32_t* __restrict__ bucket_start, int tr, int rot, int NA, int nD,
                                                         const int32_t* __restrict__ bids, const float4* __restrict__ spos, const float4* __restrict__ snrm,
                                                         int nB, uint2* __restrict__ ranges, uint32_t* __restrict__ n_ranges, uint32_t* __restrict__ totals) {
    __shared__ uint32_t ss[128], se[128];
    const int b = blockIdx.x, list = blockIdx.y, lane = threadIdx.x;
    const int i0 = bids[4 * b + 2 * list], i1 = bids[4 * b + 2 * list + 1];
    int K[4];
    ppf_compute(ld3c(spos, i0), ld3c(snrm, i0), ld3c(spos, i1), ld3c(snrm, i1), tr, rot, K);
    const bool valid = !(K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) && !(K[0] % tr || K[1] % rot || K[2] % rot || K[3] % rot);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = lane + 64 * h;                              // probe number in visiting order: a, then F1, F2, F3 ascending
        const int a = p >> 6, bi = (p >> 4) & 3, ci = (p >> 2) & 3, di = p & 3;
        const int F0 = K[0] + a * tr, F1 = K[1] + (bi - 1) * rot, F2 = K[2] + (ci - 1) * rot, F3 = K[3] + (di - 1) * rot;
        uint32_t s0 = 0, e0 = 0;
        if (valid && F1 >= 0 && F2 >= 0 && F3 >= 0) {
            const int fd = F0 / tr, f1 = F1 / rot, f2 = F2 / rot, f3 = F3 / rot;
            if (fd < nD && f1 < NA && f2 < NA && f3 < NA) {
                const uint32_t key = ppf_pack(fd, f1, f2, f3, NA);
                s0 = bucket_start[key]; e0 = bucket_start[key + 1];
                if (e0 <= s0) { s0 = 0; e0 = 0; }
            }
        }
        ss[p] = s0; se[p] = e0;
    }
    __syncthreads();
    if (lane == 0) {
        uint2* out = ranges + ((size_t)list * nB + b) * 128;
        uint32_t n = 0, total = 0, cs = 0, ce = 0;
        for (int p = 0; p < 128; ++p) {
            const uint32_t s0 = ss[p], e0 = se[p];
            if (e0 <= s0) continue;
            total += e0 - s0;
            if (n && ce == s0) ce = e0;
            else { if (n) out[n - 1] = make_uint2(cs, ce); cs = s0; ce = e0; ++n; }
        }
        if (n) out[n - 1] = make_uint2(cs, ce);
        n_ranges[(size_t)list * nB + b] = n;
        totals[(size_t)list * nB + b] = total;
    }
}

// exclusive scan of a[0 .. n) by one workgroup of 1024, in place; emit(i, offset, value) sees every element
template <class F>
__device__ __forceinline__ void block_scan_1024(uint32_t* __restrict__ a, uint32_t n, uint32_t* s_part, F emit) {
    const uint32_t tid = threadIdx.x, chunk = (n + 1023u) / 1024u;
    const uint32_t lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += a[i];
    __syncthreads();
    s_part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = a[i]; a[i] = run; emit(i, run, v); run += v; }
}

// One workgroup of 1024: offsets of every base in the gathered lists (a base without P pairs or without Q pairs gets neither,
// stocs.cpp:788), the segment arrays the gather walks, the list offsets patched into the base jobs, the totals for the host.
// Four in-place scans over arrays of nB + 1 words in device memory (any number of bases: a trial batch brings thousands), then
// everything else in parallel over the bases.
#define PLAN_MAX_BASES (1 << 20)
__global__ __launch_bounds__(1024) void plan_offsets_kernel(int nB, const uint2* __restrict__ ranges, const uint32_t* __restrict__ n_ranges, const uint32_t* __restrict__ totals,
                                                            BaseJob* __restrict__ jobs, Segment* __restrict__ psegs, Segment* __restrict__ qsegs,
                                                            uint32_t* __restrict__ p_off, uint32_t* __restrict__ q_off, uint32_t* __restrict__ sp_off, uint32_t* __restrict__ sq_off,
                                                            PlanOut* __restrict__ out, unsigned int* __restrict__ err) {
    __shared__ uint32_t s_part[1024];
    __shared__ unsigned long long s_tot[2];
    if (threadIdx.x < 2) s_tot[threadIdx.x] = 0ull;
    if (threadIdx.x < 64) err[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long tp = 0, tq = 0;
    for (int b = threadIdx.x; b <= nB; b += blockDim.x) {
        uint32_t np = 0, nq = 0, rp = 0, rq = 0;
        if (b < nB) { np = totals[b]; nq = totals[nB + b]; rp = n_ranges[b]; rq = n_ranges[nB + b]; }
        if (np == 0 || nq == 0) { np = 0; nq = 0; rp = 0; rq = 0; }
        p_off[b] = np; q_off[b] = nq; sp_off[b] = rp; sq_off[b] = rq;      // counts now, exclusive offsets after the scans (element nB: the totals)
        tp += np; tq += nq;
    }
    for (int off = 32; off > 0; off >>= 1) { tp += __shfl_xor(tp, off, 64); tq += __shfl_xor(tq, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_tot[0], tp); atomicAdd(&s_tot[1], tq); }
    __syncthreads();
    block_scan_1024(p_off, (uint32_t)nB + 1u, s_part, [&](uint32_t b, uint32_t o, uint32_t v) { if (b < (uint32_t)nB) { jobs[b].p_off = o; jobs[b].p_len = v; } });
    __syncthreads();
    block_scan_1024(q_off, (uint32_t)nB + 1u, s_part, [&](uint32_t b, uint32_t o, uint32_t v) { if (b < (uint32_t)nB) { jobs[b].q_off = o; jobs[b].q_len = v; } });
    __syncthreads();
    block_scan_1024(sp_off, (uint32_t)nB + 1u, s_part, [](uint32_t, uint32_t, uint32_t) {});
    __syncthreads();
    block_scan_1024(sq_off, (uint32_t)nB + 1u, s_part, [](uint32_t, uint32_t, uint32_t) {});
    __syncthreads();
    if (threadIdx.x == 0) {
        out->totP = s_tot[0]; out->totQ = s_tot[1]; out->n_pseg = sp_off[nB]; out->n_qseg = sq_off[nB];
        out->overflow = (s_tot[0] >= 0xFFFF0000ull || s_tot[1] >= 0xFFFF0000ull) ? 1u : 0u; out->pad = 0;
    }
}

// The segment arrays the gathers walk, from the offsets laid out above: one wavefront per (base, list) -- the <= 128 ranges of a
// lookup, two per lane, their destinations an exclusive wavefront scan of their lengths.  (Inside plan_offsets_kernel, one thread
// per base, this loop was 170 us for the 6 000 bases of a 64-trial batch.)
__global__ __launch_bounds__(256) void plan_segments_kernel(int nB, const uint2* __restrict__ ranges, const uint32_t* __restrict__ n_ranges, const uint32_t* __restrict__ totals,
                                                            Segment* __restrict__ psegs, Segment* __restrict__ qsegs, const uint32_t* __restrict__ p_off,
                                                            const uint32_t* __restrict__ q_off, const uint32_t* __restrict__ sp_off, const uint32_t* __restrict__ sq_off) {
    const int lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);       // (base, list)
    if (job >= 2 * nB) return;
    const int b = job >> 1, list = job & 1;
    if (totals[b] == 0 || totals[nB + b] == 0) return;          // a base without P pairs or without Q pairs gets neither (stocs.cpp:788)
    const uint2* r = ranges + ((size_t)list * nB + b) * 128;
    Segment* sg = list ? qsegs + sq_off[b] : psegs + sp_off[b];
    const uint32_t d0 = list ? q_off[b] : p_off[b];
    const uint32_t n = n_ranges[(size_t)list * nB + b];
    uint32_t carry = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        const uint2 v = k < n ? r[k] : make_uint2(0u, 0u);
        const uint32_t len = v.y - v.x;
        uint32_t inc = len;
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = __shfl_up(inc, dd, 64); if (lane >= dd) inc += o; }
        if (k < n) { Segment o = {v.x, len, d0 + carry + (inc - len), (uint32_t)b}; sg[k] = o; }
        carry += __shfl(inc, 63, 64);
    }
}

// normalset.h:97-104 + utils.h:139-148: int truncation of coord/epsilon, x fastest
// (cell = 1 / eg with eg a power of two, normalset.h:117-119: dividing by it and multiplying by eg are the same float)
__device__ __forceinline__ int64_t index_pos(V3 p, float cell, int eg) {
    (void)cell;
    const V3 cp = p * (float)eg;
    return (int64_t)(int)cp.z * eg * eg + ((int64_t)(int)cp.y * eg + (int64_t)(int)cp.x);
}

// Pair lists of all bases out of the device index (a lookup is the union of <= 128 buckets, ppf_index.hip), each entry
// with its sort key (base << cell_bits | position cell).  P entries sit at p1 + inv1 (p2 - p1) (nset.addElement,
// stocs.cpp:810-818), Q entries query at p1 + inv2 (p2 - p1) (stocs.cpp:827-836).  A cell the table cannot hold -- it
// cannot occur for points of the unit cube -- gets the all-ones cell: never queried, never matched.
#define GATHER_EPT 4
#define GATHER_TILE (256u * GATHER_EPT)
// (8 192 since round 5b: with 2 048 -- one round of resident workgroups, each with an eighth of a per cent of the list -- the last of them ran on a
//  half-empty chip: 4.4 wavefronts per SIMD on average over a 42-trial piece; four times as many, shorter workgroups level that: congruent phase -2 %)
static inline unsigned gather_max_wgs() { static const unsigned v = getenv("STOCS_GATHER_WGS") ? (unsigned)std::max(1, atoi(getenv("STOCS_GATHER_WGS"))) : 8192u; return v; }
#define GATHER_MAX_WGS gather_max_wgs()
static inline unsigned gather_grid(unsigned long long total) { const unsigned long long t = (total + GATHER_TILE - 1) / GATHER_TILE; return (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(t, GATHER_MAX_WGS)); }
// po != NULL: the launch was sized by a CAPACITY (the host has not read the plan yet, see stocs_internal_find_congruent): the
// list's length and segment count are the planned ones, read here, and the workgroups beyond them leave at once.
template <class KeyT>
__device__ __forceinline__ void gather_key_body(const uint32_t* __restrict__ pairs, const Segment* __restrict__ segs, int nseg, uint32_t total,
                                                const BaseJob* __restrict__ jobs, const float4* __restrict__ munit, int is_q, int cell_bits,
                                                long long cell_limit, KeyT* __restrict__ keys, uint32_t* __restrict__ vals,
                                                uint32_t* __restrict__ occ, const PlanOut* __restrict__ po, uint32_t lds_words, uint32_t n_bases) {
    // occ: ONE BIT per (base, cell) value of the key -- which (base, cell) this list occupies.  lds_words > 0 (= 2^cell_bits / 32, the
    // words of one base): the bits of the base the workgroup's current tile starts in are collected in LDS and OR-ed into the table
    // when the workgroup moves on to the next base or ends (a base's stretch is ~88 tiles at Cm: one atomic per word and workgroup
    // instead of one scattered store per entry); entries of another base in the same tile (a boundary tile) go to the table directly.
    extern __shared__ uint32_t s_occ[];
    if (po) {
        if (po->overflow) return;          // >= 2^32 planned entries: the segments' 32-bit destinations have wrapped; the host reports STOCS_ERR_CAPACITY
        const unsigned long long t = is_q ? po->totQ : po->totP;
        total = t < (unsigned long long)total ? (uint32_t)t : total;      // never beyond the buffers (a plan beyond the capacity is redone by the host)
        nseg = (int)(is_q ? po->n_qseg : po->n_pseg);
    }
    // A workgroup takes a contiguous run of tiles (GATHER_EPT * 256 entries each; a thread GATHER_EPT of them 256 apart, coalesced),
    // the launch at most GATHER_MAX_WGS workgroups: the binary search over the segments (12 scalar loads one after the other) runs once
    // per workgroup, the segment of a wavefront's entries is wave-uniform STATE carried from tile to tile (a segment holds thousands
    // of entries: the next boundary is compared, not loaded), and the workgroup owns the LDS bits of "its" base across its tiles.  The
    // steady state is pair load -> two model points -> stores: three dependent round trips per tile, ~10 us per tile with eight
    // workgroups on a CU (device clock of one workgroup, round 4).  The kernel is bound by the bytes its waves keep in flight (1.5 TB/s
    // over both lists, waves waiting 53 %, no unit above 0.65 busy); by ablation (one list, 9.5 M entries, 103 us): without the
    // occupancy marks 83, without the (key, pair) stores 83, with nothing but the pair load 62.  Prefetching the next tile's pairs,
    // the model in LDS and 64-bit products as shifts were measured and not kept.
    const uint32_t n_tiles = (total + GATHER_TILE - 1u) / GATHER_TILE;
    const uint32_t per_wg = (n_tiles + gridDim.x - 1u) / gridDim.x;
    const uint32_t t_begin = blockIdx.x * per_wg, t_end = min(n_tiles, t_begin + per_wg);
    if (t_begin >= t_end) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t w0 = t_begin * GATHER_TILE + wave * 64u;      // the wavefront's first entry
    const bool mark_lds = occ != NULL && lds_words != 0u;
    if (mark_lds) { for (uint32_t w = threadIdx.x; w < lds_words; w += blockDim.x) s_occ[w] = 0u; __syncthreads(); }
    // the base the current TILE starts in, followed through the base jobs (uniform over the workgroup: it depends on the tile alone)
    uint32_t cur_b = 0, cur_end = 0;
    if (mark_lds) {
        const uint32_t ts = t_begin * GATHER_TILE;
        int blo = 0, bhi = (int)n_bases - 1;       // last base whose stretch begins at or before ts
        while (blo < bhi) {
            const int mid = (blo + bhi + 1) >> 1;
            const uint32_t off = is_q ? jobs[mid].q_off : jobs[mid].p_off;
            if (off <= ts) blo = mid; else bhi = mid - 1;
        }
        cur_b = (uint32_t)blo;
        cur_end = blo + 1 < (int)n_bases ? (is_q ? jobs[blo + 1].q_off : jobs[blo + 1].p_off) : 0xFFFFFFFFu;
    }
    auto flush = [&](uint32_t b) {
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < lds_words; w += blockDim.x) {
            const uint32_t v = s_occ[w];
            if (v) { atomicOr(&occ[(size_t)b * lds_words + w], v); s_occ[w] = 0u; }
        }
        __syncthreads();
    };
    int lo = 0, hi = nseg - 1;                                    // last segment with dst <= w0 (uniform binary search, scalar loads)
    if (w0 < total) {
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (segs[mid].dst <= w0) lo = mid; else hi = mid - 1;
        }
    }
    // wave-uniform segment state
    int sgu = lo;
    uint32_t sg_src = 0, sg_dst = 0, sg_base = 0, next_dst = 0xFFFFFFFFu, gu = 1;
    float ju = 0.0f;
    auto load_segment = [&](int i) {
        const Segment sg = segs[i];
        sg_src = __builtin_amdgcn_readfirstlane(sg.src); sg_dst = __builtin_amdgcn_readfirstlane(sg.dst); sg_base = __builtin_amdgcn_readfirstlane(sg.base);
        next_dst = i + 1 < nseg ? __builtin_amdgcn_readfirstlane(segs[i + 1].dst) : 0xFFFFFFFFu;
        const BaseJob& J = jobs[sg_base];
        ju = is_q ? J.inv2 : J.inv1; gu = (uint32_t)J.egSize;
    };
    if (nseg > 0) load_segment(sgu);
    const KeyT cmask = ((KeyT)1 << cell_bits) - (KeyT)1;
    for (uint32_t tile = t_begin; tile < t_end; ++tile) {
        if (mark_lds && tile * GATHER_TILE >= cur_end) {            // (uniform over the workgroup)
            flush(cur_b);
            while (tile * GATHER_TILE >= cur_end) { ++cur_b; cur_end = cur_b + 1 < n_bases ? (is_q ? jobs[cur_b + 1].q_off : jobs[cur_b + 1].p_off) : 0xFFFFFFFFu; }
        }
        uint32_t e[GATHER_EPT], pr[GATHER_EPT], base[GATHER_EPT];
        bool live[GATHER_EPT];
        float inv[GATHER_EPT]; int eg[GATHER_EPT];
#pragma unroll
        for (int k = 0; k < GATHER_EPT; ++k) {
            const uint32_t ef = tile * GATHER_TILE + (uint32_t)k * 256u + wave * 64u;   // (uniform)
            e[k] = ef + lane;
            live[k] = e[k] < total;
            pr[k] = 0; base[k] = 0; inv[k] = 0.0f; eg[k] = 1;
            if (ef >= total) continue;
            const uint32_t el = min(ef + 63u, total - 1u);
            while (next_dst <= ef) load_segment(++sgu);
            if (el < next_dst) {                    // the 64 entries lie in one segment: nothing but the pair load
                if (live[k]) { pr[k] = pairs[sg_src + (e[k] - sg_dst)]; base[k] = sg_base; inv[k] = ju; eg[k] = (int)gu; }
            } else if (live[k]) {                   // a wavefront across a boundary: per-lane walk from the uniform segment
                int sgi = sgu;
                while (sgi + 1 < nseg && segs[sgi + 1].dst <= e[k]) ++sgi;
                const Segment sg = segs[sgi];
                pr[k] = pairs[sg.src + (e[k] - sg.dst)];
                base[k] = sg.base;
                const BaseJob& J = jobs[sg.base];
                inv[k] = is_q ? J.inv2 : J.inv1; eg[k] = J.egSize;
            }
        }
        float4 a1[GATHER_EPT], a2[GATHER_EPT];
#pragma unroll
        for (int k = 0; k < GATHER_EPT; ++k) { a1[k] = munit[pr[k] >> 16]; a2[k] = munit[pr[k] & 0xFFFF]; }
#pragma unroll
        for (int k = 0; k < GATHER_EPT; ++k) {
            if (!live[k]) continue;
            const V3 p1 = mk3(a1[k].x, a1[k].y, a1[k].z), p2 = mk3(a2[k].x, a2[k].y, a2[k].z);
            const int64_t pc = index_pos(p1 + inv[k] * (p2 - p1), 0.0f, eg[k]);
            const bool no_cell = pc < 0 || pc >= cell_limit;
            const KeyT key = ((KeyT)base[k] << cell_bits) | (no_cell ? cmask : (KeyT)pc);
            keys[e[k]] = key;
            vals[e[k]] = pr[k];
            if (occ && !no_cell) {
                if (mark_lds && base[k] == cur_b) atomicOr(&s_occ[(uint32_t)pc >> 5], 1u << ((uint32_t)pc & 31u));
                else atomicOr(&occ[(size_t)(key >> 5)], 1u << ((uint32_t)key & 31u));
            }
        }
    }
    if (mark_lds) flush(cur_b);
}
template <class KeyT>
__global__ __launch_bounds__(256) void gather_key_kernel(const uint32_t* __restrict__ pairs, const Segment* __restrict__ segs, int nseg, uint32_t total,
                                                         const BaseJob* __restrict__ jobs, const float4* __restrict__ munit, int is_q, int cell_bits,
                                                         long long cell_limit, KeyT* __restrict__ keys, uint32_t* __restrict__ vals,
                                                         uint32_t* __restrict__ occ, const PlanOut* __restrict__ po, uint32_t lds_words, uint32_t n_bases) {
    gather_key_body<KeyT>(pairs, segs, nseg, total, jobs, munit, is_q, cell_bits, cell_limit, keys, vals, occ, po, lds_words, n_bases);
}
// Both lists in ONE launch (blockIdx.y: 0 = P, 1 = Q): the one-stream form of count_pass -- P and Q still share the chip, and no event edge
// between two streams (~11 us each on this runtime, five of them in a trial) stands between the steps.
template <class KeyT>
struct GatherSide { const Segment* segs; int nseg; uint32_t total; KeyT* keys; uint32_t* vals; uint32_t* occ; };
template <class KeyT>
__global__ __launch_bounds__(256) void gather_key_dual_kernel(const uint32_t* __restrict__ pairs, GatherSide<KeyT> P, GatherSide<KeyT> Q, const BaseJob* __restrict__ jobs,
                                                              const float4* __restrict__ munit, int cell_bits, long long cell_limit, const PlanOut* __restrict__ po,
                                                              uint32_t lds_words, uint32_t n_bases) {
    const bool q = blockIdx.y == 1;
    gather_key_body<KeyT>(pairs, q ? Q.segs : P.segs, q ? Q.nseg : P.nseg, q ? Q.total : P.total, jobs, munit, q ? 1 : 0, cell_bits, cell_limit, q ? Q.keys : P.keys,
                          q ? Q.vals : P.vals, q ? Q.occ : P.occ, po, lds_words, n_bases);
}

// ---- survivors ----
// A P entry can only ever be matched by a Q entry of the same (base, position cell) and the other way round (Q9: only the
// query's own cell is inspected), and on the metric workload three entries out of four have no partner cell at all (CPU census
// of a Cm trial: 23 % of 7.9 M Q entries and 27 % of 10 M P entries do).  So each gather also marks the cells its list occupies,
// and both lists are reduced to the entries whose cell the OTHER list occupies -- in gather order, so the stable sorts that
// follow see the same relative order -- before anything is sorted: the sorts, the records and the join work on a quarter of
// the entries.  Entries dropped here have an empty partner run: they contribute no quad and no rank of the walk order.
#define SURV_TILE 1024
template <class KeyT>
__device__ __forceinline__ bool occ_test(const uint32_t* __restrict__ occ, KeyT key) { return (occ[(size_t)(key >> 5)] >> ((uint32_t)key & 31u)) & 1u; }

// per tile of SURV_TILE entries: how many survive.  A workgroup takes a run of consecutive GROUPS of four tiles, wavefront w of it the w-th
// tile of the group: its 16 rows of 64 keys are requested together (the round-4 form took one tile per trip of the whole workgroup -- four
// loads per thread in flight between two barriers, ~10 us per trip: 0.4 TB/s over a 40-trial piece), its count is its own (no reduction
// across wavefronts), its 16 ballots leave as one 128-byte store.  lds_words > 0: the OTHER list's bits of the base that holds the group's
// middle entry sit in LDS (2^cell_bits / 8 bytes, loaded when that base changes -- one barrier per group, two more on a change), so the
// test of an entry of that base is an LDS read -- the byte table of rounds 3-4a was gathered from device memory per entry and missed the
// L2 half of the time (3.2 MB per list and trial against a 4 MB L2 that the lists stream through).  Entries of another base read the
// table directly.
#define SURV_ROWS (SURV_TILE / 64)
template <class KeyT>
__device__ __forceinline__ void survivors_count_body(const KeyT* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ other,
                                                     uint32_t* __restrict__ tile_cnt, const PlanOut* __restrict__ po, int is_q,
                                                     unsigned long long* __restrict__ alive_bits, uint32_t lds_words, int cell_bits) {
    extern __shared__ uint32_t s_occ[];
    __shared__ KeyT s_mid[2];
    if (po) { if (po->overflow) return; const unsigned long long t = is_q ? po->totQ : po->totP; n = t < (unsigned long long)n ? (uint32_t)t : n; }   // (overflow: the gather wrote nothing)
    const uint32_t n_tiles = (n + SURV_TILE - 1u) / SURV_TILE, n_groups = (n_tiles + 3u) / 4u;
    const uint32_t per_wg = (n_groups + gridDim.x - 1u) / gridDim.x;
    const uint32_t g_begin = blockIdx.x * per_wg, g_end = min(n_groups, g_begin + per_wg);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const KeyT cmask = ((KeyT)1 << cell_bits) - (KeyT)1;
    KeyT cur = ~(KeyT)0;                 // base whose bits are in LDS
    for (uint32_t g = g_begin; g < g_end; ++g) {
        const uint32_t tile = g * 4u + w, e0 = tile * SURV_TILE + lane;       // (n < 2^32 - 2^16: no wrap; a tile beyond the list loads nothing)
        KeyT key[SURV_ROWS];
#pragma unroll
        for (int k = 0; k < SURV_ROWS; ++k) { const uint32_t e = e0 + (uint32_t)k * 64u; key[k] = (tile < n_tiles && e < n) ? keys[e] : ~(KeyT)0; }
        if (lds_words) {
            // the group's middle entry is row 0, lane 0 of wavefront 2; a group that ends before it takes its first entry (wavefront 0)
            const bool has_mid = g * 4u * SURV_TILE + 2u * SURV_TILE < n;
            if (lane == 0u && w == (has_mid ? 2u : 0u)) s_mid[g & 1u] = key[0] >> cell_bits;
            __syncthreads();                                       // (also: every wavefront is done with the bits of the group before)
            const KeyT tb = s_mid[g & 1u];
            if (tb != cur) {                                       // (uniform)
                cur = tb;
                for (uint32_t i = threadIdx.x; i < lds_words; i += blockDim.x) s_occ[i] = other[(size_t)cur * lds_words + i];
                __syncthreads();
            }
        }
        if (tile >= n_tiles) continue;                             // (wave-uniform; the barriers above are behind it)
        uint32_t cnt = 0;
        unsigned long long mine = 0ull;
#pragma unroll
        for (int k = 0; k < SURV_ROWS; ++k) {
            const uint32_t e = e0 + (uint32_t)k * 64u;
            bool ob;
            if (e >= n) ob = false;
            else if (lds_words && (key[k] >> cell_bits) == cur) { const uint32_t cl = (uint32_t)(key[k] & cmask); ob = (s_occ[cl >> 5] >> (cl & 31u)) & 1u; }
            else ob = occ_test(other, key[k]);
            const unsigned long long am = __ballot(ob);            // (the all-ones cell is never marked)
            // one bit per entry, kept for the compaction: it then reads 8 bytes per row instead of testing again,
            // and loads the keys and pairs of the survivors only (a quarter of the entries)
            if (lane == (uint32_t)k) mine = am;
            cnt += (uint32_t)__popcll(am);
        }
        if (lane < (uint32_t)SURV_ROWS) alive_bits[(size_t)tile * SURV_ROWS + lane] = mine;
        if (lane == 0u) tile_cnt[tile] = cnt;
    }
}
template <class KeyT>
__global__ __launch_bounds__(256) void survivors_count_kernel(const KeyT* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ other,
                                                              uint32_t* __restrict__ tile_cnt, const PlanOut* __restrict__ po, int is_q,
                                                              unsigned long long* __restrict__ alive_bits, uint32_t lds_words, int cell_bits) {
    survivors_count_body<KeyT>(keys, n, other, tile_cnt, po, is_q, alive_bits, lds_words, cell_bits);
}
template <class KeyT>
struct SurvSide { const KeyT* keys; const uint32_t* vals; uint32_t n; const uint32_t* other; uint32_t* tiles; unsigned long long* alive; KeyT* okeys; uint32_t* ovals; };
template <class KeyT>
__global__ __launch_bounds__(256) void survivors_count_dual_kernel(SurvSide<KeyT> P, SurvSide<KeyT> Q, const PlanOut* __restrict__ po, uint32_t lds_words, int cell_bits) {
    const bool q = blockIdx.y == 1;
    survivors_count_body<KeyT>(q ? Q.keys : P.keys, q ? Q.n : P.n, q ? Q.other : P.other, q ? Q.tiles : P.tiles, po, q ? 1 : 0, q ? Q.alive : P.alive, lds_words, cell_bits);
}

// workgroup 0: P list, workgroup 1: Q list.  Tile counts -> tile offsets (element n_tiles receives the total)
__global__ __launch_bounds__(1024) void survivors_scan_kernel(uint32_t* __restrict__ tiles_p, uint32_t ntp, uint32_t* __restrict__ tiles_q, uint32_t ntq) {
    __shared__ uint32_t s_part[1024];
    const bool q = blockIdx.x == 1;
    block_scan_1024(q ? tiles_q : tiles_p, (q ? ntq : ntp) + 1u, s_part, [](uint32_t, uint32_t, uint32_t) {});
}

// The same scan for long lists (a trial batch: 10^5 tiles), on as many workgroups as it takes -- the single workgroup above walks 137 counts
// per thread, one at a time, for a 16-trial piece (137 us on one CU, 1.4 % of the batch).  Two launches: every workgroup scans its 4 096
// counts (coalesced loads into LDS, 16 per thread there) and writes their sum; then every workgroup adds the sums of the workgroups in
// front of it (a few dozen words, reduced by the workgroup itself) to its counts.  blockIdx.y: 0 = P list, 1 = Q list.
#define TSCAN 4096
__global__ __launch_bounds__(256) void tile_scan_local_kernel(uint32_t* __restrict__ tiles_p, uint32_t np, uint32_t* __restrict__ tiles_q, uint32_t nq, uint32_t* __restrict__ part,
                                                              uint32_t parts_p) {
    __shared__ uint32_t s_v[TSCAN + 16];
    __shared__ uint32_t s_w[4];
    const bool q = blockIdx.y == 1;
    uint32_t* a = q ? tiles_q : tiles_p;
    const uint32_t n = q ? nq : np;
    const uint32_t base = blockIdx.x * TSCAN;
    if (base >= n) return;
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
#pragma unroll
    for (int j = 0; j < TSCAN / 256; ++j) { const uint32_t i = base + (uint32_t)j * 256u + t; s_v[(uint32_t)j * 256u + t] = i < n ? a[i] : 0u; }
    __syncthreads();
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { v[j] = s_v[t * 16u + (uint32_t)j]; sum += v[j]; }
    uint32_t inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += x; }
    if (lane == 63u) s_w[w] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t x = 0; x < w; ++x) run += s_w[x];
#pragma unroll
    for (int j = 0; j < 16; ++j) { s_v[t * 16u + (uint32_t)j] = run; run += v[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TSCAN / 256; ++j) { const uint32_t i = base + (uint32_t)j * 256u + t; if (i < n) a[i] = s_v[(uint32_t)j * 256u + t]; }
    if (t == 255u) part[(q ? parts_p : 0u) + blockIdx.x] = run;
}
__global__ __launch_bounds__(256) void tile_scan_add_kernel(uint32_t* __restrict__ tiles_p, uint32_t np, uint32_t* __restrict__ tiles_q, uint32_t nq, const uint32_t* __restrict__ part,
                                                            uint32_t parts_p) {
    __shared__ uint32_t s_w[4];
    const bool q = blockIdx.y == 1;
    uint32_t* a = q ? tiles_q : tiles_p;
    const uint32_t n = q ? nq : np;
    const uint32_t base = blockIdx.x * TSCAN;
    if (base >= n || blockIdx.x == 0) return;
    const uint32_t* mine = part + (q ? parts_p : 0u);
    uint32_t acc = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256u) acc += mine[j];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    const uint32_t off = s_w[0] + s_w[1] + s_w[2] + s_w[3];
#pragma unroll
    for (int j = 0; j < TSCAN / 256; ++j) { const uint32_t i = base + (uint32_t)j * 256u + threadIdx.x; if (i < n) a[i] += off; }
}

// Where every base's stretch begins and ends in the reduced lists (the lists are base-major, so that is the number of
// survivors in front of the stretch's old bounds: the offset of the tile a bound falls into plus the survivors of that
// tile in front of it), patched into the base jobs and the offset arrays the join reads.  Workgroup (b, list).
template <class KeyT>
__global__ __launch_bounds__(256) void survivors_base_offsets_kernel(const KeyT* __restrict__ pkeys, uint32_t nP, const uint32_t* __restrict__ occ_q,
                                                                     const uint32_t* __restrict__ tiles_p, const KeyT* __restrict__ qkeys, uint32_t nQ,
                                                                     const uint32_t* __restrict__ occ_p, const uint32_t* __restrict__ tiles_q, int nB,
                                                                     BaseJob* __restrict__ jobs, uint32_t* __restrict__ p_off, uint32_t* __restrict__ q_off,
                                                                     const PlanOut* __restrict__ po) {
    __shared__ uint32_t s_w[2][4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool q = blockIdx.y == 1;
    // po != NULL: nP / nQ are the CAPACITIES the key buffers and the tile arrays were sized by, the base jobs carry the PLANNED
    // stretches.  A plan beyond a capacity is redone by the host with exact sizes (count_pass returns 1 at its read-back), so this
    // launch has nothing to deliver -- and must not follow planned offsets past the gathered keys and the scanned tiles (the
    // unguarded form of round 4 read keys[e] beyond d_pk_raw / d_qk_raw and used what it found there as an occupancy index, up to
    // 512 MB past the table).  The test is uniform over the grid: either every workgroup works or none does.
    if (po && (po->overflow || po->totP > (unsigned long long)nP || po->totQ > (unsigned long long)nQ)) return;
    const KeyT* keys = q ? qkeys : pkeys;
    const uint32_t* other = q ? occ_p : occ_q;
    const uint32_t* tiles = q ? tiles_q : tiles_p;
    const uint32_t cap = q ? nQ : nP;
    uint32_t r0 = q ? jobs[b].q_off : jobs[b].p_off, r1 = r0 + (q ? jobs[b].q_len : jobs[b].p_len);
    r0 = min(r0, cap); r1 = min(max(r1, r0), cap);      // (belt and braces: tiles[] holds cap / SURV_TILE + 1 entries, keys[] cap)
    uint32_t got[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t r = h ? r1 : r0, t0 = (r / SURV_TILE) * SURV_TILE;
        uint32_t cnt = 0;
        for (uint32_t e = t0 + threadIdx.x; e < r; e += 256) cnt += occ_test(other, keys[e]) ? 1u : 0u;
        for (int d = 32; d; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if (lane == 0) s_w[h][w] = cnt;
        got[h] = tiles[r / SURV_TILE];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n0 = got[0] + s_w[0][0] + s_w[0][1] + s_w[0][2] + s_w[0][3], n1 = got[1] + s_w[1][0] + s_w[1][1] + s_w[1][2] + s_w[1][3];
        if (q) { jobs[b].q_off = n0; jobs[b].q_len = n1 - n0; q_off[b] = n0; if (b == nB - 1) q_off[nB] = n1; }
        else   { jobs[b].p_off = n0; jobs[b].p_len = n1 - n0; p_off[b] = n0; if (b == nB - 1) { p_off[nB] = n1; q_off[nB + 1] = n1; } }   // q_off[nB + 1]: P's total rides along with the Q offsets
    }
}

// the survivors of a tile, in order, behind the tile's offset: one WAVEFRONT per tile (four tiles per workgroup), no LDS, no barrier -- the
// tile's 16 ballots arrive as one 128-byte load and are handed out through scalar registers, the 16 rows of the survivors' keys and pairs
// are requested together (the round-4 form: one workgroup per tile, four rows per thread, 75 000 workgroups for a 40-trial piece)
template <class KeyT>
__device__ __forceinline__ void survivors_compact_body(const KeyT* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                       const unsigned long long* __restrict__ alive_bits, const uint32_t* __restrict__ tile_off,
                                                       KeyT* __restrict__ okeys, uint32_t* __restrict__ ovals, const PlanOut* __restrict__ po, int is_q) {
    if (po) { if (po->overflow) return; const unsigned long long t = is_q ? po->totQ : po->totP; n = t < (unsigned long long)n ? (uint32_t)t : n; }
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4u + w;
    if ((unsigned long long)tile * SURV_TILE >= (unsigned long long)n) return;      // (wave-uniform)
    const unsigned long long mine = lane < (uint32_t)SURV_ROWS ? alive_bits[(size_t)tile * SURV_ROWS + lane] : 0ull;   // the count pass's ballots
    uint32_t base = tile_off[tile];
    const uint32_t e0 = tile * SURV_TILE + lane;
    KeyT key[SURV_ROWS];
    uint32_t val[SURV_ROWS];
    unsigned long long bm[SURV_ROWS];
#pragma unroll
    for (int k = 0; k < SURV_ROWS; ++k) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, k), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), k);
        bm[k] = ((unsigned long long)hi << 32) | lo;
        key[k] = (KeyT)0; val[k] = 0u;
        if ((bm[k] >> lane) & 1ull) { key[k] = keys[e0 + (uint32_t)k * 64u]; val[k] = vals[e0 + (uint32_t)k * 64u]; }       // survivors only (a set bit is an entry below n)
    }
#pragma unroll
    for (int k = 0; k < SURV_ROWS; ++k) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm[k], 0u));
        if ((bm[k] >> lane) & 1ull) { okeys[base + rank] = key[k]; ovals[base + rank] = val[k]; }
        base += (uint32_t)__popcll(bm[k]);
    }
}
template <class KeyT>
__global__ __launch_bounds__(256) void survivors_compact_kernel(const KeyT* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                                const unsigned long long* __restrict__ alive_bits, const uint32_t* __restrict__ tile_off,
                                                                KeyT* __restrict__ okeys, uint32_t* __restrict__ ovals, const PlanOut* __restrict__ po, int is_q) {
    survivors_compact_body<KeyT>(keys, vals, n, alive_bits, tile_off, okeys, ovals, po, is_q);
}
// Both lists in one launch, into ONE list: P's survivors, then Q's right behind them (q_off[nB + 1] = P's total, written by
// survivors_base_offsets_kernel) -- so that ONE segmented sort over 2 nB segments sorts both (blockIdx.y == 2: its segment offsets, P's bases then
// Q's shifted by P's total).
template <class KeyT>
__global__ __launch_bounds__(256) void survivors_compact_dual_kernel(SurvSide<KeyT> P, SurvSide<KeyT> Q, const PlanOut* __restrict__ po, const uint32_t* __restrict__ p_off,
                                                                     const uint32_t* __restrict__ q_off, int nB, uint32_t* __restrict__ comb_off) {
    // (a plan beyond the capacities is redone by the host: survivors_base_offsets_kernel wrote nothing then, q_off[nB + 1] is not P's total)
    if (po && (po->overflow || po->totP > (unsigned long long)P.n || po->totQ > (unsigned long long)Q.n)) return;
    if (blockIdx.y == 2) {
        const uint32_t totP = q_off[nB + 1];
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i <= 2u * (uint32_t)nB; i += gridDim.x * 256u) comb_off[i] = i < (uint32_t)nB ? p_off[i] : totP + q_off[i - (uint32_t)nB];
        return;
    }
    const bool q = blockIdx.y == 1;                                // (ONE inlined body: two of them under an if / else took 248 VGPRs)
    const uint32_t shift = q ? q_off[nB + 1] : 0u;
    survivors_compact_body<KeyT>(q ? Q.keys : P.keys, q ? Q.vals : P.vals, q ? Q.n : P.n, q ? Q.alive : P.alive, q ? Q.tiles : P.tiles, (q ? Q.okeys : P.okeys) + shift,
                                 (q ? Q.ovals : P.ovals) + shift, po, q ? 1 : 0);
}

// Zero fill as an ordinary kernel on the context's stream
__global__ __launch_bounds__(256) void zero_u32_kernel(uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ b) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0; if (b) b[i] = 0; }
}

// After the sort: one record per P entry -- invPoint of stocs.cpp:845-849 (once per P entry instead of once per (Q, P)
// test) and the direction cell of normalset.hpp:114-131 (0xFFFF: never inserted) -- and for every (base, position cell)
// the run of its P entries.  Replaces the pointer grid _grid[pId] -> AngularGrid of normalset.h:87-88.
template <class KeyT>
__global__ __launch_bounds__(256) void p_records_kernel(const KeyT* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t totP, int cell_bits,
                                                        long long NC, const BaseJob* __restrict__ jobs, const float4* __restrict__ munit,
                                                        const float4* __restrict__ mpos, float nepsilon, float4* __restrict__ prec,
                                                        uint16_t* __restrict__ pdc, uint32_t* __restrict__ cfirst, uint32_t* __restrict__ cend) {
    // four entries per thread, 256 apart: the four (key, pair) loads, then the model points of all four, are in flight together (one entry per
    // thread left the kernel waiting on three dependent round trips: 1.8 TB/s over the 66 M entries of a 32-trial piece).
    // prec == NULL (the distance gate cannot fail inside a position cell, CongruentState::close_cells): the join decides on direction cells
    // alone, nobody reads the world-space point -- neither it nor the two model positions behind it are touched.
    constexpr int R = 4;
    const uint32_t e0 = blockIdx.x * (256u * R) + threadIdx.x;
    const KeyT cmask = ((KeyT)1 << cell_bits) - (KeyT)1;
    KeyT key[R], kprev[R], knext[R];
    uint32_t pr[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint32_t e = e0 + (uint32_t)k * 256u;
        const bool in = e < totP;
        key[k] = in ? keys[e] : (KeyT)0; pr[k] = in ? vals[e] : 0u;
        kprev[k] = (cfirst && in && e > 0u) ? keys[e - 1u] : (KeyT)0;
        knext[k] = (cfirst && in && e + 1u < totP) ? keys[e + 1u] : (KeyT)0;
    }
    float4 ua[R], ub[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { ua[k] = munit[pr[k] >> 16]; ub[k] = munit[pr[k] & 0xFFFF]; }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint32_t e = e0 + (uint32_t)k * 256u;
        if (e >= totP) continue;
        const uint32_t b = (uint32_t)(key[k] >> cell_bits);
        const KeyT pc = key[k] & cmask;
        int dc = 0xFFFF;
        if (pc != cmask) {
            const int nc = index_normal(normalized3(mk3(ub[k].x, ub[k].y, ub[k].z) - mk3(ua[k].x, ua[k].y, ua[k].z)), nepsilon);
            if (nc >= 0 && nc < 343) dc = nc;
        }
        if (prec) {
            const int ia = pr[k] >> 16, ib = pr[k] & 0xFFFF;
            const V3 pp1 = ld3c(mpos, ia), pp2 = ld3c(mpos, ib);
            const V3 ip = pp1 + (pp2 - pp1) * jobs[b].inv1;
            prec[e] = make_float4(ip.x, ip.y, ip.z, __int_as_float(dc));
        }
        if (pdc) pdc[e] = (uint16_t)dc;
        if (!cfirst || pc == cmask) continue;
        const bool first = (e == 0) || kprev[k] != key[k];
        const bool last = (e + 1 == totP) || knext[k] != key[k];
        if (first) cfirst[(long long)b * NC + (long long)pc] = e;
        if (last) cend[(long long)b * NC + (long long)pc] = e + 1;
    }
}

// Everything the join needs, by value.
template <class KeyT>
struct JoinArgs {
    const BaseJob* jobs; const uint32_t* q_off; int nB;
    const float4* munit; const float4* mpos;
    const KeyT* qkeys; const uint32_t* qvals; uint32_t totQ;   // Q entries in (base, cell, index position) order
    const KeyT* pkeys; const uint32_t* pvals; const float4* prec;
    const uint16_t* pdc;   // direction cells alone when the distance gate of stocs.cpp:854 cannot fail inside a position cell (else NULL)
    const uint32_t* cfirst; const uint32_t* cend; long long NC;
    float nepsilon, half_inv_neps, dist_thr;
    const float2* trig;    // [STOCS_MAX_CONE + 1][STOCS_MAX_CONE]: (cos, sin) of theta_a for every sample count
    int id_bits, cell_bits;
    int base_in_key;   // the packed quad key carries the base above the four ids (it does whenever 4 id_bits + base_bits <= 64)
#ifdef STOCS_TOOLS_BUILD
    int gmin, rmin;   // STOCS_JOIN_GMIN / STOCS_JOIN_RMIN: the thresholds of join_count_kernel's group-wise counting, for sweeps
    int ablate;   // measurement build (STOCS_JOIN_ABLATE): 1 = no cone sampling (every direction cell set), 2 = no walk over the P run
#endif
};
#ifdef STOCS_TOOLS_BUILD
#define STOCS_JOIN_ABLATE(A, bit) (((A).ablate & (bit)) != 0)
#define JOIN_GMIN(A) (A).gmin
#define JOIN_RMIN(A) (uint32_t)(A).rmin
#else
#define STOCS_JOIN_ABLATE(A, bit) false
#define JOIN_GMIN(A) JOIN_GROUP_MIN
#define JOIN_RMIN(A) (uint32_t)JOIN_RUN_MIN
#endif

// the run of P entries that live in position cell `key` (only the query's own cell is inspected, Q9)
template <class KeyT>
__device__ __forceinline__ void p_run(const JoinArgs<KeyT>& A, const BaseJob& J, uint32_t b, KeyT key, KeyT pc, uint32_t* lo, uint32_t* hi) {
    if (A.cfirst) {
        *lo = A.cfirst[(long long)b * A.NC + (long long)pc];
        *hi = A.cend[(long long)b * A.NC + (long long)pc];
        return;
    }
    const KeyT* keys = A.pkeys + J.p_off;
    uint32_t l = 0, h = J.p_len;
    while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] < key) l = mid + 1; else h = mid; }
    *lo = J.p_off + l;
    h = J.p_len;
    while (l < h) { const uint32_t mid = (l + h) >> 1; if (keys[mid] <= key) l = mid + 1; else h = mid; }
    *hi = J.p_off + l;
}

// Cone tables of the bases a workgroup's Q entries belong to, staged in LDS: the (x, y) components of a base's samples
// (z = cos alpha is folded into the filter).  A workgroup's 256 consecutive entries of the (base, cell)-sorted list
// belong to one base, seldom two; entries of a base beyond JOIN_LDS_BASES read the table from global memory.
#define JOIN_LDS_BASES 2
struct JoinLds {
    uint32_t seen[256][11];                          // 343-bit direction-cell set per lane (11 is odd: no bank conflicts)
    float2 dirs[JOIN_LDS_BASES][STOCS_MAX_CONE];
};

// The join of ONE Q entry against the P entries of its position cell: stocs.cpp:827-858 + normalset.hpp:166-214.
//   MODE 0: count;  MODE 1: write every match to out[0..] (walk order).
// b0 = base of the workgroup's first entry (its table is lds.dirs[0], the next base's lds.dirs[1]).
//   MODE 2: count, but only up to the direction-cell set: the P run comes back in *run_lo / *run_hi (empty: nothing can match)
//           and the caller counts (join_count_kernel does that per (base, cell) group of its workgroup).
template <int MODE, class KeyT>
__device__ __forceinline__ uint32_t join_one(const JoinArgs<KeyT>& A, uint32_t i, JoinLds& lds, uint32_t b0, uint64_t* __restrict__ out,
                                             uint32_t* run_lo = NULL, uint32_t* run_hi = NULL) {
    uint32_t* my = lds.seen[threadIdx.x];
    const KeyT key = A.qkeys[i];
    const KeyT cmask = ((KeyT)1 << A.cell_bits) - (KeyT)1;
    const KeyT pc = key & cmask;
    if (pc == cmask) return 0;
    const uint32_t b = (uint32_t)(key >> A.cell_bits);
    const BaseJob& J = A.jobs[b];
    const int nb = J.nb;
    if (J.p_len == 0 || nb == 0) return 0;
    uint32_t lo, hi;
    p_run(A, J, b, key, pc, &lo, &hi);
    if (lo >= hi) return 0;
    const uint32_t qr = A.qvals[i];
    const int qa = qr >> 16, qb = qr & 0xFFFF;
    const V3 p1 = ld3c(A.munit, qa), p2 = ld3c(A.munit, qb);
    const V3 pq1 = ld3c(A.mpos, qa), pq2 = ld3c(A.mpos, qb);
    const V3 queryQ = pq1 + J.inv2 * (pq2 - pq1);
    const V3 queryn = normalized3(p2 - p1);
    // direction cells hit by the sampled cone (std::set<unsigned> colored of normalset.hpp:188-204)
#pragma unroll
    for (int k = 0; k < 11; ++k) my[k] = 0;
    float q[4];
    quat_from_z(queryn, q);
    const float dz = J.cos_alpha;
    const ConeFilter cf = cone_filter_setup(q, dz, A.half_inv_neps);
    auto colour = [&](float dx, float dy) {
        int id = cone_cell_filtered(cf, dx, dy);
        if (id < 0) {   // within 5e-5 of a cell boundary (or not finite): the reference's own arithmetic decides
            id = cone_cell_exact(q, mk3(dx, dy, dz), A.nepsilon);
            if (id < 0) return;  // std::array::at would throw (NaN direction)
        }
        // this lane's own words; the no-return LDS atomic is one instruction and needs no wait
        __hip_atomic_fetch_or(&my[id >> 5], 1u << (id & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    if (STOCS_JOIN_ABLATE(A, 1)) {
#pragma unroll
        for (int k = 0; k < 11; ++k) my[k] = 0xFFFFFFFFu;
    } else
    if (b - b0 < (uint32_t)JOIN_LDS_BASES) {   // the sample after this one is read while this one is evaluated
        const float2* tab = lds.dirs[b - b0];
        float2 d = tab[0];
        for (int a = 0; a < nb; ++a) {
            const float2 dn = tab[a + 1 < nb ? a + 1 : a];
            colour(d.x, d.y);
            d = dn;
        }
    } else {
        const float2* tr = A.trig + (size_t)nb * STOCS_MAX_CONE;
        const float sa = J.sin_alpha;
        for (int a = 0; a < nb; ++a) { const float2 t = tr[a]; colour(sa * t.x, sa * t.y); }
    }
    // one linear pass over the position cell's P entries against the direction bitset, four records in flight
    uint32_t local = 0;
    const int id_bits = A.id_bits;
    auto test = [&](const float4& r, uint32_t k) {
        const uint32_t dc = (uint32_t)__float_as_int(r.w);
        if (dc >= 343u || !((my[dc >> 5] >> (dc & 31)) & 1u)) return;
        if (sqn3(queryQ - mk3(r.x, r.y, r.z)) <= A.dist_thr) {  // squared metres vs metres (Q1), reproduced
            if (MODE == 1) {   // sort key: base, then (P.first, P.second, Q.first, Q.second) == the std::set order
                const uint32_t pr = A.pvals[k];
                const int pa = pr >> 16, pb = pr & 0xFFFF;
                out[local] = (A.base_in_key ? (uint64_t)b << (4 * id_bits) : 0ull) | ((uint64_t)pa << (3 * id_bits)) | ((uint64_t)pb << (2 * id_bits)) |
                             ((uint64_t)qa << id_bits) | (uint64_t)qb;
            }
            local++;
        }
    };
    if (MODE == 2) { *run_lo = lo; *run_hi = hi; return 0; }
    if (STOCS_JOIN_ABLATE(A, 2)) return my[0] & 1u;
    if (MODE == 0 && A.pdc) {
        // Counting with the gate out of the way: ||e_Q - e_P||^2 <= epsilon (squared metres against metres, Q1) holds for ANY
        // two points of one position cell -- the cell edge is below 2 epsilon, so the squared diagonal is below 12 epsilon^2,
        // and the host enables this path only when that is comfortably below epsilon -- so an entry matches iff its direction
        // cell is in the cone's set: 2 bytes per entry instead of 16, eight entries per load.
        const uint16_t* dcs = A.pdc;
        auto hit = [&](uint32_t dc) { return dc < 343u && ((my[dc >> 5] >> (dc & 31)) & 1u) ? 1u : 0u; };
        uint32_t k = lo;
        for (; k < hi && (k & 7u); ++k) local += hit(dcs[k]);
        for (; k + 8 <= hi; k += 8) {
            const uint4 v = *(const uint4*)(dcs + k);
            local += hit(v.x & 0xFFFFu) + hit(v.x >> 16) + hit(v.y & 0xFFFFu) + hit(v.y >> 16) + hit(v.z & 0xFFFFu) + hit(v.z >> 16) + hit(v.w & 0xFFFFu) + hit(v.w >> 16);
        }
        for (; k < hi; ++k) local += hit(dcs[k]);
        return local;
    }
    if (A.pdc) {   // (MODE 1 with the gate out of the way: the same decision as the count above, the pairs of the matches read on demand)
        for (uint32_t k = lo; k < hi; ++k) {
            const uint32_t dc = A.pdc[k];
            if (dc >= 343u || !((my[dc >> 5] >> (dc & 31)) & 1u)) continue;
            const uint32_t pr = A.pvals[k];
            const int pa = pr >> 16, pb = pr & 0xFFFF;
            out[local++] = (A.base_in_key ? (uint64_t)b << (4 * id_bits) : 0ull) | ((uint64_t)pa << (3 * id_bits)) | ((uint64_t)pb << (2 * id_bits)) |
                           ((uint64_t)qa << id_bits) | (uint64_t)qb;
        }
        return local;
    }
    uint32_t k = lo;
    for (; k + 4 <= hi; k += 4) {
        const float4 r0 = A.prec[k], r1 = A.prec[k + 1], r2 = A.prec[k + 2], r3 = A.prec[k + 3];
        test(r0, k); test(r1, k + 1); test(r2, k + 2); test(r3, k + 3);
    }
    for (; k < hi; ++k) test(A.prec[k], k);
    return local;
}

// workgroup prologue of the join kernels: cone tables of the first JOIN_LDS_BASES bases of the workgroup's entries
template <class KeyT>
__device__ __forceinline__ uint32_t join_stage_tables(const JoinArgs<KeyT>& A, JoinLds& lds, uint32_t i0 = 0xFFFFFFFFu) {
    if (i0 == 0xFFFFFFFFu) i0 = blockIdx.x * blockDim.x;
    const uint32_t b0 = (uint32_t)(A.qkeys[i0] >> A.cell_bits);   // i0 < totQ for every launched workgroup
    for (int t = threadIdx.x; t < JOIN_LDS_BASES * STOCS_MAX_CONE; t += blockDim.x) {
        const uint32_t b = b0 + (uint32_t)(t / STOCS_MAX_CONE);
        const int a = t % STOCS_MAX_CONE;
        float2 d = make_float2(0.f, 0.f);
        if (b < (uint32_t)A.nB && a < A.jobs[b].nb) { const float2 tr = A.trig[(size_t)A.jobs[b].nb * STOCS_MAX_CONE + a]; const float sa = A.jobs[b].sin_alpha; d = make_float2(sa * tr.x, sa * tr.y); }
        lds.dirs[t / STOCS_MAX_CONE][a] = d;
    }
    __syncthreads();
    return b0;
}

// per-base totals: the scanned count at the first Q entry of every base (one small copy instead of one per base)
__global__ __launch_bounds__(256) void base_offsets_kernel(const unsigned long long* __restrict__ qoffe, const uint32_t* __restrict__ q_off, int n,
                                                           unsigned long long* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) out[b] = qoffe[q_off[b]];
}

// count pass, one lane per Q entry in (base, position cell) order
// The Q entries are in (base, cell) order, so the 64 of a wavefront fall into a few groups that walk the SAME P run, each
// entry against its own direction-cell set.  Three quarters of the entries have no P entry in their cell and are done at once;
// the rest sit in the big bases, where a cell holds ~300 Q entries and ~300 P entries, and their walks were 60 % of this kernel
// (device clock 0.32 ms: 0.14 without the walk, 0.25 without the cone samples; tools/join_ablate.py; a CPU census of a Cm trial:
// 18 % of the entries carry 98 % of the walk).  When the distance gate cannot fail (A.pdc: direction cells alone decide) a
// match count is the sum over the set's cells of (P entries of the run in that cell): a group of >= JOIN_GROUP_MIN lanes whose
// run has >= JOIN_RUN_MIN entries builds that histogram once per wavefront in LDS (16-bit counters, the wavefront's 64 lanes
// reading the run two bytes per entry) and every lane of the group adds up <= 56 counters instead of testing ~300 entries.
// Smaller groups and shorter runs walk the run as before.  No workgroup barrier: wavefronts without work leave at once.
// Same counts either way (integer sums).
#define JOIN_GROUP_MIN 1
#define JOIN_RUN_MIN 64
template <class KeyT>
__global__ __launch_bounds__(256) void join_count_kernel(JoinArgs<KeyT> A, unsigned long long* __restrict__ qcnt) {
    __shared__ JoinLds lds;
    __shared__ uint32_t s_hist[4][172];            // per wavefront: 344 16-bit counters, two per word
    const uint32_t b0 = join_stage_tables(A, lds);
    const uint32_t tid = threadIdx.x, i = blockIdx.x * blockDim.x + tid;
    const int lane = tid & 63, w = tid >> 6;
    const bool live = i < A.totQ;
    if (i == 0) qcnt[A.totQ] = 0;   // the scan runs over totQ + 1 entries so that its last output is the total
    if (!A.pdc || STOCS_JOIN_ABLATE(A, 3)) {   // the gate has to be evaluated per (Q, P) couple: every entry walks its run
        if (live) qcnt[i] = join_one<0>(A, i, lds, b0, (uint64_t*)NULL);
        return;
    }
    uint32_t lo = 0, hi = 0;
    if (live) (void)join_one<2>(A, i, lds, b0, (uint64_t*)NULL, &lo, &hi);   // direction-cell set in lds.seen[tid], the run in lo / hi
    const bool has = hi > lo;
    if (!__any(has)) { if (live) qcnt[i] = 0; return; }
    // groups of the wavefront = maximal stretches of lanes with the same run
    const uint32_t plo = __shfl_up(lo, 1, 64), phi = __shfl_up(hi, 1, 64);
    const bool head = lane == 0 || plo != lo || phi != hi;
    const unsigned long long heads = __ballot(head);
    const int g0 = 63 - __clzll((long long)(heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull))));
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int g1 = above ? lane + __ffsll((long long)above) : 64;
    const bool big = has && (hi - lo) < 65536u && (g1 - g0) >= JOIN_GMIN(A) && (hi - lo) >= JOIN_RMIN(A);
    const uint32_t* my = lds.seen[tid];
    const uint16_t* dcs = A.pdc;
    uint32_t count = 0;
    if (has && !big) {   // small groups / short runs: walk the run, eight 2-byte direction cells per load
        auto hit = [&](uint32_t dc) { return dc < 343u && ((my[dc >> 5] >> (dc & 31)) & 1u) ? 1u : 0u; };
        uint32_t k = lo;
        for (; k < hi && (k & 7u); ++k) count += hit(dcs[k]);
        for (; k + 8 <= hi; k += 8) {
            const uint4 v = *(const uint4*)(dcs + k);
            count += hit(v.x & 0xFFFFu) + hit(v.x >> 16) + hit(v.y & 0xFFFFu) + hit(v.y >> 16) + hit(v.z & 0xFFFFu) + hit(v.z >> 16) + hit(v.w & 0xFFFFu) + hit(v.w >> 16);
        }
        for (; k < hi; ++k) count += hit(dcs[k]);
    }
    // the big groups of this wavefront, one after the other (at most 64 / JOIN_GROUP_MIN of them)
    unsigned long long todo = __ballot(head && big);
    uint32_t* hist = s_hist[w];
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t glo = (uint32_t)__shfl((int)lo, leader, 64), ghi = (uint32_t)__shfl((int)hi, leader, 64);
        const int gend = __shfl(g1, leader, 64);
        for (int c = lane; c < 172; c += 64) hist[c] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k0 = glo; k0 < ghi; k0 += 256) {   // four loads in flight per lane
            uint32_t dc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + 64u * u + (uint32_t)lane; dc[u] = k < ghi ? dcs[k] : 0xFFFFu; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (dc[u] < 343u) atomicAdd(&hist[dc[u] >> 1], 1u << (16u * (dc[u] & 1u)));
        }
        __builtin_amdgcn_wave_barrier();
        if (lane >= leader && lane < gend) {
#pragma unroll
            for (int ww = 0; ww < 11; ++ww) {
                uint32_t bits = my[ww];
                while (bits) {
                    const uint32_t dc = 32u * ww + (uint32_t)(__ffs((int)bits) - 1);
                    count += (hist[dc >> 1] >> (16u * (dc & 1u))) & 0xFFFFu;
                    bits &= bits - 1u;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (live) qcnt[i] = count;
}

// fill pass for the bases whose out_base is not ~0: destinations from the exclusive scan of the counts (no atomics).
// `blocks` lists, per workgroup, the (first, end) Q entries it takes: only the Q ranges of the selected bases are walked
// (a trial materialises a handful of small bases out of a hundred: the sweep over all Q entries was 50 us at Cm).
template <class KeyT>
__global__ __launch_bounds__(256) void join_fill_kernel(JoinArgs<KeyT> A, const unsigned long long* __restrict__ qoffe,
                                                        const unsigned long long* __restrict__ out_base, uint64_t* __restrict__ quads,
                                                        const uint2* __restrict__ blocks) {
    __shared__ JoinLds lds;
    const uint2 range = blocks[blockIdx.x];
    const uint32_t b0 = join_stage_tables(A, lds, range.x);
    const uint32_t i = range.x + threadIdx.x;
    if (i >= range.y) return;
    const uint32_t b = (uint32_t)(A.qkeys[i] >> A.cell_bits);
    const unsigned long long ob = out_base[b];
    if (ob == ~0ull) return;
    const unsigned long long o0 = qoffe[i];
    if (qoffe[i + 1] == o0) return;
    join_one<1>(A, i, lds, b0, quads + ob + (o0 - qoffe[A.q_off[b]]));
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
// reference's std::set order) or a rank in a big base's WALK order (Q entries in (cell, index) order, each one's matches
// in P-run order): the Q entry is found by binary search in the scanned counts and its join is re-run up to
// the wanted match, so the 10^7-10^8 quads of the big bases are never materialised.  One wavefront per pick:
// the cone samples are spread over the lanes (LDS bitset, the reference's exact arithmetic), the P run is tested 64
// entries at a time and the wanted match is located with ballot / popcount, in run order.
template <class KeyT>
__global__ __launch_bounds__(256) void resolve_picks_kernel(JoinArgs<KeyT> A, const unsigned long long* __restrict__ qoffe, const Pick* __restrict__ picks, int n,
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
        uint32_t elo = q0, ehi = q1 - 1;   // last i in [q0, q1) with qoffe[i] <= target
        while (elo < ehi) {
            const uint32_t mid = elo + ((ehi - elo + 1) >> 1);
            if (qoffe[mid] <= target) elo = mid; else ehi = mid - 1;
        }
        unsigned long long want = target - qoffe[elo];
        // ---- the join of Q entry elo, wave-wide (same arithmetic as join_one) ----
        const BaseJob& J = A.jobs[b];
        const KeyT qk = A.qkeys[elo];
        const KeyT cmask = ((KeyT)1 << A.cell_bits) - (KeyT)1;
        const KeyT pc = qk & cmask;
        const uint32_t qr = A.qvals[elo];
        const int qa = qr >> 16, qb = qr & 0xFFFF;
        const V3 p1 = ld3c(A.munit, qa), p2 = ld3c(A.munit, qb);
        const V3 pq1 = ld3c(A.mpos, qa), pq2 = ld3c(A.mpos, qb);
        const V3 queryQ = pq1 + J.inv2 * (pq2 - pq1);
        const V3 queryn = normalized3(p2 - p1);
        uint32_t lo = 0, hi = 0;
        // a Q entry without a position cell has no matches and is never selected; stay in bounds regardless
        if (pc != cmask) p_run(A, J, (uint32_t)b, qk, pc, &lo, &hi);
        if (lane < 12) seen[lane] = 0;
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
        float q[4];
        quat_from_z(queryn, q);
        if (lane < J.nb) {
            const float2 tr = A.trig[(size_t)J.nb * STOCS_MAX_CONE + lane];
            const int id = cone_cell_exact(q, mk3(J.sin_alpha * tr.x, J.sin_alpha * tr.y, J.cos_alpha), A.nepsilon);
            if (id >= 0) atomicOr(&seen[id >> 5], 1u << (id & 31));
        }
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
        const int id_bits = A.id_bits;
        for (uint32_t k0 = lo; k0 < hi; k0 += 64) {
            const uint32_t k = k0 + lane;
            bool hit = false;
            int pa = 0, pb = 0;
            if (k < hi && A.pdc) {   // the gate cannot fail inside a position cell: direction cells alone (no records were written)
                const uint32_t dc = A.pdc[k];
                if (dc < 343u && ((seen[dc >> 5] >> (dc & 31)) & 1u)) {
                    const uint32_t pr = A.pvals[k];
                    pa = pr >> 16; pb = pr & 0xFFFF;
                    hit = true;
                }
            } else if (k < hi) {
                const float4 r = A.prec[k];
                const uint32_t dc = (uint32_t)__float_as_int(r.w);
                if (dc < 343u && ((seen[dc >> 5] >> (dc & 31)) & 1u)) {
                    const uint32_t pr = A.pvals[k];
                    pa = pr >> 16; pb = pr & 0xFFFF;
                    hit = sqn3(queryQ - mk3(r.x, r.y, r.z)) <= A.dist_thr;
                }
            }
            const unsigned long long m = __ballot(hit);
            const unsigned long long cnt = (unsigned long long)__popcll(m);
            if (want < cnt) {
                const unsigned long long below = m & ((1ull << lane) - 1ull);
                if (hit && (unsigned long long)__popcll(below) == want)
                    key = (A.base_in_key ? (uint64_t)b << (4 * id_bits) : 0ull) | ((uint64_t)pa << (3 * id_bits)) | ((uint64_t)pb << (2 * id_bits)) |
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
// demand: the sorted Q list, the sorted P list with its records and cell tables and the scanned per-Q match counts.
struct CongruentState {
    Arena arena_state;   // the buffers below + the temporaries of stocs_find_congruent_all; reset by that call
    Arena arena_tmp;     // temporaries of the calls that produce quads afterwards; reset by each of them
    bool valid = false;  // a count pass has completed and its buffers are intact
    // host work of the current call that only the join needs (the cone records of a batch's thousands of bases: three libm calls each,
    // 0.2 ms for 6 400 bases), done by count_pass while the device gathers instead of before anything is enqueued
    std::function<int()> deferred;
    // quads of the small bases, materialised ahead of the picks that refer to them (stocs_internal_prepare_small)
    bool small_ready = false, small_any = false;
    DevBuf<uint64_t> d_small_sorted;
    DevBuf<unsigned long long> d_small_soff;
    bool wide = false;   // 64-bit sort keys (only without the cell table: more than 32 bits of (base, cell))
    int nB = 0;
    uint32_t totP = 0, totQ = 0;
    long long NC = 0;
    bool use_table = false;
    float nepsilon = 0, half_inv_neps = 0;
    int id_bits = 16, base_bits = 1, cell_bits = 1;
    bool base_in_key = true;   // packed quad keys carry the base (4 id_bits + base_bits <= 64); else bases are told apart by their runs
    DevBuf<BaseJob> d_jobs;
    DevBuf<uint32_t> d_qoff, d_pvals, d_qvals, d_cfirst, d_cend;
    DevBuf<char> d_pkeys, d_qkeys;   // KeyT arrays (uint32_t, or uint64_t when wide)
    DevBuf<float4> d_prec;
    DevBuf<uint16_t> d_pdc;
    bool close_cells = false;     // the distance gate cannot fail inside a position cell: count on direction cells alone
    DevBuf<unsigned long long> d_qoffe;
    DevBuf<int32_t> d_bids;
    DevBuf<unsigned int> d_err;   // picks resolve_picks_kernel could not resolve (internal consistency check)
    // host sources of asynchronous uploads issued by materialise / make_jobs: kept here so that they outlive the copy
    // (those calls return without synchronising; the caller's own synchronisation point comes before the next reuse)
    std::vector<unsigned long long> h_out_base, h_off;
    std::vector<uint32_t> h_qoff;          // Q range of every base (host copy of d_qoff)
    bool reduce = false;                   // this trial's lists are reduced to the entries with a partner cell
    unsigned long long hist_P = 0, hist_Q = 0;   // planned list lengths of the last trial on this scene (capacities of the next one)
    int hist_nB = 0;
    bool no_quads = false;                 // the last count found no (base, cell) that both lists occupy
    std::vector<uint2> h_blocks;           // workgroup -> Q range of the last materialise
    void* h_stage = NULL;         // pinned staging of the per-trial tables (one upload per trial)
    size_t stage_bytes = 0;
    float2* d_trig = NULL;        // the shared (cos, sin) table of the cone samples, uploaded once per context
    char* d_plan = NULL;          // persistent planning buffer: jobs, base ids, ranges, segments, offsets (outside the arenas:
    size_t plan_bytes = 0;        // it is written before the trial's sizes -- and with them the arena's -- are known)
    template <class KeyT>
    JoinArgs<KeyT> args(const stocs_ctx* c) const {
        JoinArgs<KeyT> A;
        A.jobs = d_jobs.p; A.q_off = d_qoff.p; A.nB = nB; A.munit = c->d_munit; A.mpos = c->d_mpos;
        A.qkeys = (const KeyT*)d_qkeys.p; A.qvals = d_qvals.p; A.totQ = totQ;
        A.pkeys = (const KeyT*)d_pkeys.p; A.pvals = d_pvals.p; A.prec = d_prec.p; A.pdc = close_cells ? d_pdc.p : NULL;
        A.cfirst = use_table ? d_cfirst.p : NULL; A.cend = use_table ? d_cend.p : NULL; A.NC = NC;
        A.nepsilon = nepsilon; A.half_inv_neps = half_inv_neps; A.dist_thr = c->prm.distance_threshold; A.id_bits = id_bits; A.cell_bits = cell_bits;
        A.trig = d_trig;
        A.base_in_key = base_in_key ? 1 : 0;
#ifdef STOCS_TOOLS_BUILD
        A.gmin = JOIN_GROUP_MIN; A.rmin = JOIN_RUN_MIN;
        A.ablate = getenv("STOCS_JOIN_ABLATE") ? atoi(getenv("STOCS_JOIN_ABLATE")) : 0;
        if (getenv("STOCS_JOIN_GMIN")) A.gmin = atoi(getenv("STOCS_JOIN_GMIN"));
        if (getenv("STOCS_JOIN_RMIN")) A.rmin = atoi(getenv("STOCS_JOIN_RMIN"));
#endif
        return A;
    }
};

// Materialises the quads of the bases with sel[b] != 0 into one device buffer, sorted by (base, a, b, c, d) =
// per base the order of the reference's std::set<pair<P index, Q index>>.  off[b] .. off[b+1] is base b's run.
template <class KeyT>
static int materialise_t(stocs_ctx* c, CongruentState* S, const std::vector<char>& sel, DevBuf<uint64_t>* out, std::vector<unsigned long long>* off) {
    const int nB = S->nB;
    std::vector<unsigned long long>& out_base = S->h_out_base;
    out_base.assign(nB, ~0ull);
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
    DevBuf<unsigned long long> d_ob; DevBuf<uint64_t> d_raw; DevBuf<char> d_tmp; DevBuf<uint2> d_blocks;
    std::vector<uint2>& blocks = S->h_blocks;
    blocks.clear();
    for (int b = 0; b < nB; ++b)
        if (sel[b] && c->quad_off[b + 1] > c->quad_off[b])
            for (uint32_t i0 = S->h_qoff[b]; i0 < S->h_qoff[b + 1]; i0 += 256) blocks.push_back(make_uint2(i0, std::min(i0 + 256u, S->h_qoff[b + 1])));
    int rc;
    if ((rc = d_ob.alloc(nB)) || (rc = d_raw.alloc(tot)) || (rc = out->alloc(tot)) || (rc = d_blocks.alloc(std::max<size_t>(blocks.size(), 1)))) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(d_ob.p, out_base.data(), 8 * (size_t)nB, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(d_blocks.p, blocks.data(), sizeof(uint2) * blocks.size(), hipMemcpyHostToDevice, st));
    if (!blocks.empty())
        hipLaunchKernelGGL(join_fill_kernel<KeyT>, dim3((unsigned)blocks.size()), dim3(256), 0, st, S->args<KeyT>(c), S->d_qoffe.p, d_ob.p, d_raw.p, d_blocks.p);
    STOCS_HIP_CHECK(hipGetLastError());
    size_t tmp = 0;
    int n_sel = 0;
    for (int b = 0; b < nB; ++b) n_sel += sel[b] ? 1 : 0;
    if (S->base_in_key || n_sel <= 1) {   // one sort over everything: the base is part of the key (or there is only one)
        const unsigned end_bit = (unsigned)(4 * S->id_bits + (S->base_in_key ? S->base_bits : 0));
        STOCS_HIP_CHECK(sort_keys(NULL, tmp, d_raw.p, out->p, (size_t)tot, 0, end_bit, st));
        if ((rc = d_tmp.alloc(tmp))) return rc;
        STOCS_HIP_CHECK(sort_keys(d_tmp.p, tmp, d_raw.p, out->p, (size_t)tot, 0, end_bit, st));
    } else {
        // models beyond 16 384 points at 100 bases: four 15- or 16-bit ids fill the 64 bits, so the key holds the ids alone and
        // every base's run (off[b] .. off[b+1], filled base by base) is sorted as a segment of its own
        DevBuf<unsigned long long> d_off;
        if ((rc = d_off.alloc((size_t)nB + 1))) return rc;
        STOCS_HIP_CHECK(hipMemcpyAsync(d_off.p, off->data(), 8 * ((size_t)nB + 1), hipMemcpyHostToDevice, st));   // *off outlives the copy (S->h_off or the caller's)
        const unsigned end_bit = (unsigned)(4 * S->id_bits);
        STOCS_HIP_CHECK(segmented_sort_keys(NULL, tmp, d_raw.p, out->p, (unsigned)tot, (unsigned)nB, d_off.p, d_off.p + 1, 0, end_bit, st));
        if ((rc = d_tmp.alloc(tmp))) return rc;
        STOCS_HIP_CHECK(segmented_sort_keys(d_tmp.p, tmp, d_raw.p, out->p, (unsigned)tot, (unsigned)nB, d_off.p, d_off.p + 1, 0, end_bit, st));
    }
    // no synchronisation: the temporaries are arena memory, recycled only by a later call's reset, and every later use is
    // ordered behind this work on the context's stream (out_base was copied from pageable memory: staged by the runtime
    // before hipMemcpyAsync returned)
    return STOCS_OK;
}
static int materialise(stocs_ctx* c, CongruentState* S, const std::vector<char>& sel, DevBuf<uint64_t>* out, std::vector<unsigned long long>* off) {
    return S->wide ? materialise_t<uint64_t>(c, S, sel, out, off) : materialise_t<uint32_t>(c, S, sel, out, off);
}

// (cos, sin) of the sample angles theta_a = a * angleStep for every sample count nb (normalset.hpp:183-188: float libm on per-count
// scalars, exactly the reference's values); row nb of [STOCS_MAX_CONE + 1][STOCS_MAX_CONE]
static const float* cone_trig() {
    static float trig[STOCS_MAX_CONE + 1][STOCS_MAX_CONE][2];
    static bool ready = false;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    if (!ready) {
        memset(trig, 0, sizeof(trig));
        for (unsigned nb = 1; nb <= STOCS_MAX_CONE; ++nb) {
            const float angleStep = (float)((double)2.0f * M_PI / (double)(float)nb);
            for (unsigned a = 0; a < nb; ++a) {
                const float theta = (float)a * angleStep;
                trig[nb][a][0] = cosf(theta);
                trig[nb][a][1] = sinf(theta);
            }
        }
        ready = true;
    }
    pthread_mutex_unlock(&mu);
    return &trig[0][0][0];
}

// the cone fields of the base jobs, uploaded on their own when the host computes them while the device already plans and gathers
__global__ __launch_bounds__(256) void patch_cone_kernel(BaseJob* __restrict__ jobs, const float4* __restrict__ cone, int nB) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nB) return;
    const float4 v = cone[b];
    jobs[b].cos_alpha = v.x; jobs[b].sin_alpha = v.y; jobs[b].nb = __float_as_int(v.z);
}

// cone samples of a base: normalset.hpp:178-190 (float libm calls on per-base scalars: exactly the reference's values): their
// number and sin(alpha); sample a is (sin_alpha * cos theta_a, sin_alpha * sin theta_a, cos_alpha) with the shared table above
static void fill_cone_table(BaseJob* J) {
    const float alpha = acosf(J->cos_alpha);
    const float perimeter = (float)((double)2.0f * M_PI * (double)atanf(alpha));  // sic (Q10)
    const unsigned nb = (unsigned)(2 * ceilf(perimeter * 7.0f / 2.0f));
    J->sin_alpha = sinf(alpha);
    J->nb = (nb > STOCS_MAX_CONE || !(alpha == alpha)) ? 0 : (int)nb;  // nb <= 56 for any alpha in [0, pi]; NaN alpha -> no samples
}

static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
#define STOCS_TICK(label)                                                                          \
    if (dbg) { (void)hipStreamSynchronize(c->stream); const double t_ = now_s(); fprintf(stderr, "[stocs congruent] %-18s %8.3f ms\n", label, (t_ - tprev) * 1e3); tprev = t_; }

// device part of stocs_find_congruent_all for one key width
// The stable sort of a pair list by (base, cell).  The lists are base-major already (the gather, and the compaction of the survivors, emit
// base after base), so 32-bit keys go through the library's own SEGMENTED onesweep (sort32.hip, round 5): every base's stretch -- seg_off,
// the per-base offsets on the device -- is sorted by its cell bits alone, two passes whatever the number of bases (rocPRIM over all
// significant bits: three for a single trial's Q list, four for a 40-trial piece).  STOCS_SORT=rocprim keeps rocPRIM's radix_sort_pairs
// selectable for A/B; 64-bit keys (position grids beyond 2^32 (base, cell) values) always take it.  `own` says which one ran: the own sort
// keeps an error word at the start of its temporary block.
static bool cong_sort_own() { static const bool own = !(getenv("STOCS_SORT") && !strcmp(getenv("STOCS_SORT"), "rocprim")); return own; }
static hipError_t cong_sort(void* tmp, size_t& bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned cell_bits, unsigned end_bit,
                            const uint32_t* seg_off, int n_seg, hipStream_t st, bool* own) {
    *own = cong_sort_own() && n_seg > 0 && n < ((size_t)1 << 30);   // (30-bit prefixes in its look-back words)
#ifdef STOCS_TOOLS_BUILD
    // measurement build: STOCS_DUMP_SORT=<prefix> writes the unsorted list of every sort (header n, n_seg, cell_bits; keys; values; offsets) to
    // <prefix>_<k>.bin -- tools/sort_real.py replays them through stocs_debug_sort_pairs (real key skew, real base lengths)
    if (tmp && getenv("STOCS_DUMP_SORT") && n > 0) {
        static int k_dump = 0;
        (void)hipStreamSynchronize(st);
        std::vector<uint32_t> hk(n), hv(n), ho((size_t)n_seg + 1);
        (void)hipMemcpy(hk.data(), kin, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hv.data(), vin, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ho.data(), seg_off, ((size_t)n_seg + 1) * 4, hipMemcpyDeviceToHost);
        char path[512];
        snprintf(path, sizeof path, "%s_%d.bin", getenv("STOCS_DUMP_SORT"), k_dump++);
        if (FILE* f = fopen(path, "wb")) {
            const uint32_t hdr[4] = {(uint32_t)n, (uint32_t)n_seg, cell_bits, end_bit};
            fwrite(hdr, 4, 4, f); fwrite(hk.data(), 4, n, f); fwrite(hv.data(), 4, n, f); fwrite(ho.data(), 4, (size_t)n_seg + 1, f); fclose(f);
        }
    }
#endif
    return *own ? sort_pairs_own(tmp, bytes, kin, kout, vin, vout, n, 0, cell_bits, seg_off, (uint32_t)n_seg, st) : sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, end_bit, st);
}
static hipError_t cong_sort(void* tmp, size_t& bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned, unsigned end_bit,
                            const uint32_t*, int, hipStream_t st, bool* own) {
    *own = false;
    return sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, end_bit, st);
}

// the per-trial tables of a planned trial, all in the context's persistent planning buffer (device)
struct PlanDev {
    BaseJob* jobs; Segment* psegs; Segment* qsegs; uint32_t* q_off; uint32_t* p_off; int32_t* bids; unsigned int* err;
    int n_pseg, n_qseg;
};

// d_po != NULL ("optimistic"): the host has NOT read the plan -- S->totP / S->totQ are capacities the buffers and launches are
// sized by, the kernels read the planned totals from *d_po, and the plan's totals arrive with the survivors' in ONE read-back
// (po_pin).  Returns 1 (not an error) when the plan turned out larger than the capacities: the caller redoes the pass with the
// exact sizes it now knows.
template <class KeyT>
static int count_pass(stocs_ctx* c, CongruentState* S, const PlanDev& plan, bool dbg, double& tprev, const PlanOut* d_po = NULL, const PlanOut* po_pin = NULL) {
    const int nB = S->nB;
    const size_t totP0 = S->totP, totQ0 = S->totQ;   // the gathered lists as planned (d_po: their capacities)
    size_t totP = totP0, totQ = totQ0;               // the lists that are sorted and joined (the survivors, when the lists are reduced)
    hipStream_t st = c->stream;
    // The device's own clock of the kernel groups (HIP events between them, reported by stocs_last_call_timing as "device: ..." steps) is
    // OPT-IN since round 5b (stocs_set_option "device_clock" / STOCS_DEVICE_CLOCK=1): on this runtime every event recorded between two kernels
    // of a stream leaves the queue idle for ~5 us, and the nine of a call were 45 us of a Cm trial's 600.  The host's steps between the call's
    // own synchronisation points are always recorded.
    const bool dev_clock = c->device_clock != 0;
    // ONE stream (round 5b) for the reduced 32-bit form: P and Q go through every step in the SAME launch (blockIdx.y), the survivors of both
    // land in one list and ONE segmented sort over 2 nB segments sorts it.  The two-stream form of rounds 3-5a (P on the context's stream, Q on
    // the auxiliary one) paid ~11 us per event edge between the streams, five of them in a trial -- a tenth of a Cm trial's congruent phase;
    // it stays for 64-bit keys, unreduced lists and rocPRIM's sort, and under STOCS_CONGRUENT_TWO_STREAMS for A/B.
    // Lists beyond 10^8 entries -- the pieces of a trial batch at the metric size -- keep the two streams: there the edges are nothing and the
    // overlap of unlike kernels (P's records next to Q's sort) is worth 2 % (64 Cm trials: 2 040 against 1 985 trials/s).
    const int force_streams = getenv("STOCS_CONGRUENT_TWO_STREAMS") ? 2 : (getenv("STOCS_CONGRUENT_ONE_STREAM") ? 1 : 0);
    const bool one_stream = S->reduce && sizeof(KeyT) == 4 && cong_sort_own() && (totP0 + totQ0) < ((size_t)1 << 30) && nB < (1 << 22) && force_streams != 2 &&
                            (force_streams == 1 || totP0 + totQ0 < (size_t)100000000);
    hipStream_t sq = (c->aux_stream && !one_stream) ? c->aux_stream : st;
    DevBuf<KeyT> d_pk_raw, d_qk_raw;
    DevBuf<uint32_t> d_pv_raw, d_qv_raw;
    DevBuf<char> d_tmp;
    int rc;
    if ((rc = d_pk_raw.alloc(totP0)) || (rc = d_pv_raw.alloc(totP0)) || (rc = d_qk_raw.alloc(totQ0)) || (rc = d_qv_raw.alloc(totQ0))) return rc;
    S->d_jobs.p = plan.jobs;
    if (S->deferred && !S->reduce) { const int rd = S->deferred(); S->deferred = nullptr; if (rd) return rd; }   // (no survivors pass to hide it behind)
    const Segment* d_psegs = plan.psegs;
    const Segment* d_qsegs = plan.qsegs;
    S->d_qoff.p = plan.q_off;
    S->d_bids.p = plan.bids;
    S->d_err.p = plan.err;
    const int n_pseg = plan.n_pseg, n_qseg = plan.n_qseg;
    const PpfIndex& ix = c->index;
    StreamAudit& AU = c->audit;                       // STOCS_DEBUG_STREAMS: every two-stream step below says what it reads and writes
    const int s0 = 0, s1 = sq != st ? 1 : 0;
    const long long cell_limit = S->use_table ? S->NC : ((long long)1 << 31);
    const unsigned end_bit = (unsigned)(S->cell_bits + S->base_bits);
    const unsigned long long occ_bits = (unsigned long long)nB << S->cell_bits;   // (base, cell) values = BITS of an occupancy table
    // one base's bits in LDS (gather: collected there; count: the other list's, tested there) while they fit 32 KB
    const uint32_t lds_words = (S->cell_bits >= 5 && S->cell_bits <= 18 && !getenv("STOCS_CONGRUENT_NO_LDS_BITS")) ? (1u << (S->cell_bits - 5)) : 0u;
    const bool reduce = S->reduce;
    const KeyT* pk_in = d_pk_raw.p; const uint32_t* pv_in = d_pv_raw.p;   // what the sorts read
    const KeyT* qk_in = d_qk_raw.p; const uint32_t* qv_in = d_qv_raw.p;
    DevBuf<KeyT> d_pk_c, d_qk_c;
    DevBuf<uint32_t> d_pv_c, d_qv_c, d_surv, d_comb_off;
    bool have_surv_clock = false;
    if (reduce) {
        // one zeroed block: occupancy of P | occupancy of Q | P tile counts (+ total) | Q tile counts (+ total)
        const size_t W = (size_t)((occ_bits + 31) >> 5) + 1;   // words of one occupancy table
        const uint32_t ntp = (uint32_t)((totP0 + SURV_TILE - 1) / SURV_TILE), ntq = (uint32_t)((totQ0 + SURV_TILE - 1) / SURV_TILE);
        const size_t o_tp = 2 * W, o_tq = o_tp + ntp + 1, n_words = o_tq + ntq + 1;
        if ((rc = d_surv.alloc(n_words))) return rc;
        DevBuf<unsigned long long> d_bits_p, d_bits_q;   // one bit per gathered entry: survives (written by the count pass, read by the compaction)
        if ((rc = d_bits_p.alloc((size_t)std::max(ntp, 1u) * (SURV_TILE / 64))) || (rc = d_bits_q.alloc((size_t)std::max(ntq, 1u) * (SURV_TILE / 64)))) return rc;
        uint32_t* occ_p = d_surv.p;
        uint32_t* occ_q = d_surv.p + W;
        hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, d_surv.p, n_words, (uint32_t*)NULL);
        const uint32_t* tiles_p = d_surv.p + o_tp; const uint32_t* tiles_q = d_surv.p + o_tq;
        AU.use(s0, occ_p, true, "occupancy of P", "zero fill"); AU.use(s0, occ_q, true, "occupancy of Q", "zero fill");
        AU.use(s0, tiles_p, true, "tile counts of P", "zero fill"); AU.use(s0, tiles_q, true, "tile counts of Q", "zero fill");
        AU.use(s0, plan.jobs, true, "base jobs", "plan kernels"); AU.use(s0, plan.psegs, true, "P segments", "plan kernels"); AU.use(s0, plan.qsegs, true, "Q segments", "plan kernels");
        AU.use(s0, plan.q_off, true, "Q offsets per base", "plan kernels");
        if (dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[6], st));
        if (sq != st) {
            STOCS_HIP_CHECK(hipEventRecord(c->ev_fork, st));          // the plan upload and the zero fill are on st
            STOCS_HIP_CHECK(hipStreamWaitEvent(sq, c->ev_fork, 0));
            AU.record(c->ev_fork, s0); AU.wait(s1, c->ev_fork);
        }
        if (one_stream) {
            const GatherSide<KeyT> gp = {d_psegs, n_pseg, (uint32_t)totP0, d_pk_raw.p, d_pv_raw.p, occ_p}, gq = {d_qsegs, n_qseg, (uint32_t)totQ0, d_qk_raw.p, d_qv_raw.p, occ_q};
            hipLaunchKernelGGL(gather_key_dual_kernel<KeyT>, dim3(std::max(gather_grid(totP0), gather_grid(totQ0)), 2), dim3(256), lds_words * 4, st, ix.d_pairs, gp, gq,
                               (const BaseJob*)S->d_jobs.p, (const float4*)c->d_munit, S->cell_bits, cell_limit, d_po, lds_words, (uint32_t)nB);
        } else
        hipLaunchKernelGGL(gather_key_kernel<KeyT>, dim3(gather_grid(totQ0)), dim3(256), lds_words * 4, sq, ix.d_pairs, d_qsegs, n_qseg, (uint32_t)totQ0,
                           S->d_jobs.p, c->d_munit, 1, S->cell_bits, cell_limit, d_qk_raw.p, d_qv_raw.p, occ_q, d_po, lds_words, (uint32_t)nB);
        AU.use(s1, plan.qsegs, false, "Q segments", "gather Q"); AU.use(s1, plan.jobs, false, "base jobs", "gather Q");
        AU.use(s1, d_qk_raw.p, true, "gathered Q keys", "gather Q"); AU.use(s1, d_qv_raw.p, true, "gathered Q pairs", "gather Q"); AU.use(s1, occ_q, true, "occupancy of Q", "gather Q");
        if (sq != st || dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[8], sq));             // Q's cells are marked
        AU.record(c->ev_t[8], s1);
        if (!one_stream)
        hipLaunchKernelGGL(gather_key_kernel<KeyT>, dim3(gather_grid(totP0)), dim3(256), lds_words * 4, st, ix.d_pairs, d_psegs, n_pseg, (uint32_t)totP0,
                           S->d_jobs.p, c->d_munit, 0, S->cell_bits, cell_limit, d_pk_raw.p, d_pv_raw.p, occ_p, d_po, lds_words, (uint32_t)nB);
        AU.use(s0, plan.psegs, false, "P segments", "gather P"); AU.use(s0, plan.jobs, false, "base jobs", "gather P");
        AU.use(s0, d_pk_raw.p, true, "gathered P keys", "gather P"); AU.use(s0, d_pv_raw.p, true, "gathered P pairs", "gather P"); AU.use(s0, occ_p, true, "occupancy of P", "gather P");
        if (sq != st || dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[9], st));             // P's cells are marked
        AU.record(c->ev_t[9], s0);
        if (sq != st) { STOCS_HIP_CHECK(hipStreamWaitEvent(sq, c->ev_t[9], 0)); STOCS_HIP_CHECK(hipStreamWaitEvent(st, c->ev_t[8], 0)); AU.wait(s1, c->ev_t[9]); AU.wait(s0, c->ev_t[8]); }
        // (one stream: the compacted lists are ONE buffer, allocated here so that the launch below can name it)
        if (one_stream) { if ((rc = d_pk_c.alloc(totP0 + totQ0)) || (rc = d_pv_c.alloc(totP0 + totQ0)) || (rc = d_comb_off.alloc(2 * (size_t)nB + 1))) return rc; }
        const SurvSide<KeyT> sp = {(const KeyT*)d_pk_raw.p, (const uint32_t*)d_pv_raw.p, (uint32_t)totP0, (const uint32_t*)occ_q, d_surv.p + o_tp, d_bits_p.p, d_pk_c.p, d_pv_c.p},
                             sqd = {(const KeyT*)d_qk_raw.p, (const uint32_t*)d_qv_raw.p, (uint32_t)totQ0, (const uint32_t*)occ_p, d_surv.p + o_tq, d_bits_q.p, d_pk_c.p, d_pv_c.p};
        if (one_stream)
            hipLaunchKernelGGL(survivors_count_dual_kernel<KeyT>, dim3(std::max(1u, std::min((std::max(ntp, ntq) + 3u) / 4u, GATHER_MAX_WGS)), 2), dim3(256), lds_words * 4, st, sp, sqd, d_po,
                               lds_words, S->cell_bits);
        else
        hipLaunchKernelGGL(survivors_count_kernel<KeyT>, dim3(std::max(1u, std::min((ntq + 3u) / 4u, GATHER_MAX_WGS))), dim3(256), lds_words * 4, sq, (const KeyT*)d_qk_raw.p, (uint32_t)totQ0, (const uint32_t*)occ_p, d_surv.p + o_tq, d_po, 1, d_bits_q.p, lds_words, S->cell_bits);
        AU.use(s1, d_qk_raw.p, false, "gathered Q keys", "survivors count Q"); AU.use(s1, occ_p, false, "occupancy of P", "survivors count Q"); AU.use(s1, tiles_q, true, "tile counts of Q", "survivors count Q"); AU.use(s1, d_bits_q.p, true, "alive bits of Q", "survivors count Q");
        if (sq != st) { STOCS_HIP_CHECK(hipEventRecord(c->ev_join, sq)); AU.record(c->ev_join, s1); }
        if (!one_stream)
        hipLaunchKernelGGL(survivors_count_kernel<KeyT>, dim3(std::max(1u, std::min((ntp + 3u) / 4u, GATHER_MAX_WGS))), dim3(256), lds_words * 4, st, (const KeyT*)d_pk_raw.p, (uint32_t)totP0, (const uint32_t*)occ_q, d_surv.p + o_tp, d_po, 0, d_bits_p.p, lds_words, S->cell_bits);
        AU.use(s0, d_pk_raw.p, false, "gathered P keys", "survivors count P"); AU.use(s0, occ_q, false, "occupancy of Q", "survivors count P"); AU.use(s0, tiles_p, true, "tile counts of P", "survivors count P"); AU.use(s0, d_bits_p.p, true, "alive bits of P", "survivors count P");
        if (sq != st) { STOCS_HIP_CHECK(hipStreamWaitEvent(st, c->ev_join, 0)); AU.wait(s0, c->ev_join); }
        AU.use(s0, tiles_p, true, "tile counts of P", "tile scan"); AU.use(s0, tiles_q, true, "tile counts of Q", "tile scan");
        AU.use(s0, d_qk_raw.p, false, "gathered Q keys", "base offsets"); AU.use(s0, occ_p, false, "occupancy of P", "base offsets"); AU.use(s0, plan.jobs, true, "base jobs", "base offsets");
        AU.use(s0, plan.q_off, true, "Q offsets per base", "base offsets");
        if (std::max(ntp, ntq) + 1u <= 2u * TSCAN) {
            hipLaunchKernelGGL(survivors_scan_kernel, dim3(2), dim3(1024), 0, st, d_surv.p + o_tp, ntp, d_surv.p + o_tq, ntq);
        } else {      // long lists (trial batches): the scan on many workgroups
            const uint32_t pp = (ntp + 1u + TSCAN - 1u) / TSCAN, pq = (ntq + 1u + TSCAN - 1u) / TSCAN;
            DevBuf<uint32_t> d_part;
            if ((rc = d_part.alloc((size_t)pp + pq))) return rc;
            hipLaunchKernelGGL(tile_scan_local_kernel, dim3(std::max(pp, pq), 2), dim3(256), 0, st, d_surv.p + o_tp, ntp + 1u, d_surv.p + o_tq, ntq + 1u, d_part.p, pp);
            hipLaunchKernelGGL(tile_scan_add_kernel, dim3(std::max(pp, pq), 2), dim3(256), 0, st, d_surv.p + o_tp, ntp + 1u, d_surv.p + o_tq, ntq + 1u, (const uint32_t*)d_part.p, pp);
        }
        hipLaunchKernelGGL(survivors_base_offsets_kernel<KeyT>, dim3((unsigned)nB, 2), dim3(256), 0, st, (const KeyT*)d_pk_raw.p, (uint32_t)totP0, (const uint32_t*)occ_q,
                           (const uint32_t*)(d_surv.p + o_tp), (const KeyT*)d_qk_raw.p, (uint32_t)totQ0, (const uint32_t*)occ_p, (const uint32_t*)(d_surv.p + o_tq), nB,
                           S->d_jobs.p, plan.p_off, plan.q_off, d_po);
        STOCS_HIP_CHECK(hipGetLastError());
        // the host sizes the sorts and the join with the survivors' totals and lays the materialise blocks out with their Q offsets
        uint32_t* qoff_pin = (uint32_t*)((char*)c->h_pin + PIN_VAR + 8 * ((size_t)nB + 1));
        STOCS_HIP_CHECK(hipMemcpyAsync(qoff_pin, plan.q_off, 4 * ((size_t)nB + 2), hipMemcpyDeviceToHost, st));   // Q offsets, Q total, P total
        STOCS_HIP_CHECK(hipEventRecord(c->ev_t[7], st));
        AU.record(c->ev_t[7], s0);
        // the survivors move behind their tiles' offsets while the host waits for the totals: the compacted lists are sized by
        // the gathered ones here (the totals are what the wait is for)
        if (!one_stream) { if ((rc = d_pk_c.alloc(totP0)) || (rc = d_pv_c.alloc(totP0)) || (rc = d_qk_c.alloc(totQ0)) || (rc = d_qv_c.alloc(totQ0))) return rc; }
        if (sq != st) {
            STOCS_HIP_CHECK(hipEventRecord(c->ev_fork, st));          // tile offsets are scanned on st
            STOCS_HIP_CHECK(hipStreamWaitEvent(sq, c->ev_fork, 0));
            AU.record(c->ev_fork, s0); AU.wait(s1, c->ev_fork);
        }
        AU.use(s1, d_qk_raw.p, false, "gathered Q keys", "compact Q"); AU.use(s1, d_qv_raw.p, false, "gathered Q pairs", "compact Q"); AU.use(s1, d_bits_q.p, false, "alive bits of Q", "compact Q");
        AU.use(s1, tiles_q, false, "tile counts of Q", "compact Q"); AU.use(s1, d_qk_c.p, true, "surviving Q keys", "compact Q"); AU.use(s1, d_qv_c.p, true, "surviving Q pairs", "compact Q");
        AU.use(s0, d_pk_raw.p, false, "gathered P keys", "compact P"); AU.use(s0, d_pv_raw.p, false, "gathered P pairs", "compact P"); AU.use(s0, d_bits_p.p, false, "alive bits of P", "compact P");
        AU.use(s0, tiles_p, false, "tile counts of P", "compact P"); AU.use(s0, d_pk_c.p, true, "surviving P keys", "compact P"); AU.use(s0, d_pv_c.p, true, "surviving P pairs", "compact P");
        if (one_stream)
            hipLaunchKernelGGL(survivors_compact_dual_kernel<KeyT>, dim3(std::max(1u, (std::max(ntp, ntq) + 3u) / 4u), 3), dim3(256), 0, st, sp, sqd, d_po, (const uint32_t*)plan.p_off,
                               (const uint32_t*)plan.q_off, nB, d_comb_off.p);
        else {
        hipLaunchKernelGGL(survivors_compact_kernel<KeyT>, dim3((ntq + 3u) / 4u), dim3(256), 0, sq, (const KeyT*)d_qk_raw.p, (const uint32_t*)d_qv_raw.p, (uint32_t)totQ0,
                           (const unsigned long long*)d_bits_q.p, (const uint32_t*)(d_surv.p + o_tq), d_qk_c.p, d_qv_c.p, d_po, 1);
        hipLaunchKernelGGL(survivors_compact_kernel<KeyT>, dim3((ntp + 3u) / 4u), dim3(256), 0, st, (const KeyT*)d_pk_raw.p, (const uint32_t*)d_pv_raw.p, (uint32_t)totP0,
                           (const unsigned long long*)d_bits_p.p, (const uint32_t*)(d_surv.p + o_tp), d_pk_c.p, d_pv_c.p, d_po, 0);
        }
        STOCS_HIP_CHECK(hipGetLastError());
        c->timing[0].lap("enqueue gather + occupancy + survivor counts");
        if (S->deferred) { const int rd = S->deferred(); S->deferred = nullptr; if (rd) return rd; c->timing[0].lap("host: cone records of the bases (while the device gathers)"); }
        STOCS_HIP_CHECK(hipEventSynchronize(c->ev_t[7]));   // the read-back, not the compaction behind it
        AU.host_sync_event(c->ev_t[7]);
        c->timing[0].lap("wait for the device (survivors)");
        have_surv_clock = true;
        if (po_pin) {   // the plan's own totals came with this read-back: were the capacities enough?
            if (po_pin->overflow || po_pin->totP > (unsigned long long)totP0 || po_pin->totQ > (unsigned long long)totQ0) {
                STOCS_HIP_CHECK(hipStreamSynchronize(st));            // the compaction behind the read-back: nothing may still touch the arena
                if (sq != st) STOCS_HIP_CHECK(hipStreamSynchronize(sq));
                AU.host_sync(s0); AU.host_sync(s1);
                return 1;
            }
        }
        totP = qoff_pin[nB + 1]; totQ = qoff_pin[nB];
        memcpy(S->h_qoff.data(), qoff_pin, 4 * ((size_t)nB + 1));
        if (dbg) fprintf(stderr, "[stocs congruent] survivors: P %zu of %zu, Q %zu of %zu\n", totP, totP0, totQ, totQ0);
        S->totP = (uint32_t)totP; S->totQ = (uint32_t)totQ;
        if (totP == 0 || totQ == 0) { S->no_quads = true; return STOCS_OK; }   // no cell is shared: no quads (quad_off is all zero already)
        pk_in = d_pk_c.p; pv_in = d_pv_c.p; qk_in = d_qk_c.p; qv_in = d_qv_c.p;
        if (one_stream) { qk_in = d_pk_c.p + totP; qv_in = d_pv_c.p + totP; }   // (Q's survivors sit behind P's)
    }
    if (one_stream) {   // one sorted list: P's part, then Q's
        if ((rc = S->d_pkeys.alloc((totP + totQ) * sizeof(KeyT))) || (rc = S->d_pvals.alloc(totP + totQ))) return rc;
        S->d_qkeys.p = S->d_pkeys.p + totP * sizeof(KeyT); S->d_qvals.p = S->d_pvals.p + totP;
    } else if ((rc = S->d_pkeys.alloc(totP * sizeof(KeyT))) || (rc = S->d_pvals.alloc(totP)) || (rc = S->d_qkeys.alloc(totQ * sizeof(KeyT))) || (rc = S->d_qvals.alloc(totQ)))
        return rc;
    if ((rc = S->d_prec.alloc(S->close_cells ? 1 : totP)) || (rc = S->d_pdc.alloc(((size_t)totP + 15) & ~(size_t)7)))
        return rc;
    // The P side (sort, records) and the Q side (sort) are independent until the join: the Q side runs on
    // the context's auxiliary stream next to the P side (a radix pass of 7 M pairs moves ~1.8 TB/s: two of them share the chip)
    size_t tb1 = 0, tb2 = 0;
    // P side with the run table: sorted by position cell ALONE.  The gather emits base after base and the sort is stable, so
    // inside a cell the entries stay grouped by base, in index order inside a base: the runs of (base, cell) are contiguous
    // all the same, the table finds them wherever they are, and 15 bits are two radix passes where 22 are three.
    const unsigned end_bit_p = (S->use_table && !getenv("STOCS_CONGRUENT_P_FULLSORT")) ? (unsigned)S->cell_bits : end_bit;
    bool own_p = false, own_q = false;      // (decided per list: by its length per base)
    if (one_stream) STOCS_HIP_CHECK(cong_sort(NULL, tb1, pk_in, (KeyT*)S->d_pkeys.p, pv_in, S->d_pvals.p, totP + totQ, (unsigned)S->cell_bits, end_bit_p, d_comb_off.p, 2 * nB, st, &own_p));
    else {
    STOCS_HIP_CHECK(cong_sort(NULL, tb1, pk_in, (KeyT*)S->d_pkeys.p, pv_in, S->d_pvals.p, totP, (unsigned)S->cell_bits, end_bit_p, plan.p_off, nB, st, &own_p));
    STOCS_HIP_CHECK(cong_sort(NULL, tb2, qk_in, (KeyT*)S->d_qkeys.p, qv_in, S->d_qvals.p, totQ, (unsigned)S->cell_bits, end_bit, plan.q_off, nB, st, &own_q));
    }
    DevBuf<char> d_tmp2;
    if ((rc = d_tmp.alloc(tb1)) || (rc = d_tmp2.alloc(tb2))) return rc;
    // device-side clock of the groups below (HIP events on the streams they run on; read after the call's closing
    // synchronisation): a call that takes 70 ms instead of 1 then says which group of kernels it spent them in
    if (dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[0], st));
    if (sq != st) {
        STOCS_HIP_CHECK(hipEventRecord(c->ev_fork, st));          // the upload above is on st
        STOCS_HIP_CHECK(hipStreamWaitEvent(sq, c->ev_fork, 0));
        AU.record(c->ev_fork, s0); AU.wait(s1, c->ev_fork);
    }
    // The P side first: it is the longer chain (sort + records: ~100 us at Cm against ~70 us of the Q sort), and whichever side is enqueued second
    // starts ~25 us later -- the host needs that long for the first side's five launches (kernel trace of a trial, round 5)
    if (!reduce)
        hipLaunchKernelGGL(gather_key_kernel<KeyT>, dim3(gather_grid(totP)), dim3(256), 0, st, ix.d_pairs, d_psegs, n_pseg, (uint32_t)totP,
                           S->d_jobs.p, c->d_munit, 0, S->cell_bits, cell_limit, d_pk_raw.p, d_pv_raw.p, (uint32_t*)NULL, (const PlanOut*)NULL, 0u, (uint32_t)nB);
    STOCS_HIP_CHECK(hipGetLastError());
    // one stable sort per list: (base, position cell); inside a cell the entries keep the index order of the gather
    if (one_stream) STOCS_HIP_CHECK(cong_sort(d_tmp.p, tb1, pk_in, (KeyT*)S->d_pkeys.p, pv_in, S->d_pvals.p, totP + totQ, (unsigned)S->cell_bits, end_bit_p, d_comb_off.p, 2 * nB, st, &own_p));
    else
    STOCS_HIP_CHECK(cong_sort(d_tmp.p, tb1, pk_in, (KeyT*)S->d_pkeys.p, pv_in, S->d_pvals.p, totP, (unsigned)S->cell_bits, end_bit_p, plan.p_off, nB, st, &own_p));
    if (dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[2], st));
    if (S->use_table) {
        const size_t ncell = (size_t)(S->NC * nB);
        if ((rc = S->d_cfirst.alloc(ncell)) || (rc = S->d_cend.alloc(ncell))) return rc;
        // the table is read at the cells of the Q entries only: in the reduced lists every such cell holds P entries, so the
        // records kernel writes every slot that is read and the 8 bytes per (base, cell) need no zero fill
        if (!reduce) hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, S->d_cfirst.p, ncell, S->d_cend.p);
    }
    hipLaunchKernelGGL(p_records_kernel<KeyT>, dim3((unsigned)((totP + 1023) / 1024)), dim3(256), 0, st, (const KeyT*)S->d_pkeys.p, S->d_pvals.p, (uint32_t)totP,
                       S->cell_bits, S->NC, S->d_jobs.p, c->d_munit, c->d_mpos, S->nepsilon, S->close_cells ? (float4*)NULL : S->d_prec.p, S->close_cells ? S->d_pdc.p : (uint16_t*)NULL,
                       S->use_table ? S->d_cfirst.p : (uint32_t*)NULL,
                       S->use_table ? S->d_cend.p : (uint32_t*)NULL);
    STOCS_HIP_CHECK(hipGetLastError());
    AU.use(s0, pk_in, false, "P keys to sort", "sort P"); AU.use(s0, pv_in, false, "P pairs to sort", "sort P");
    AU.use(s0, S->d_pkeys.p, true, "sorted P keys", "sort P + records"); AU.use(s0, S->d_pvals.p, true, "sorted P pairs", "sort P + records"); AU.use(s0, plan.jobs, false, "base jobs", "P records");
    if (!reduce)
        hipLaunchKernelGGL(gather_key_kernel<KeyT>, dim3(gather_grid(totQ)), dim3(256), 0, sq, ix.d_pairs, d_qsegs, n_qseg, (uint32_t)totQ,
                           S->d_jobs.p, c->d_munit, 1, S->cell_bits, cell_limit, d_qk_raw.p, d_qv_raw.p, (uint32_t*)NULL, (const PlanOut*)NULL, 0u, (uint32_t)nB);
    if (!one_stream)
    STOCS_HIP_CHECK(cong_sort(d_tmp2.p, tb2, qk_in, (KeyT*)S->d_qkeys.p, qv_in, S->d_qvals.p, totQ, (unsigned)S->cell_bits, end_bit, plan.q_off, nB, sq, &own_q));
    AU.use(s1, plan.q_off, false, "Q offsets per base", "sort Q");
    AU.use(s1, qk_in, false, "Q keys to sort", "sort Q"); AU.use(s1, qv_in, false, "Q pairs to sort", "sort Q");
    AU.use(s1, S->d_qkeys.p, true, "sorted Q keys", "sort Q"); AU.use(s1, S->d_qvals.p, true, "sorted Q pairs", "sort Q"); AU.use(s1, d_tmp2.p, true, "sort scratch Q", "sort Q");
    if (dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[1], sq));
    if (sq != st) { STOCS_HIP_CHECK(hipEventRecord(c->ev_join, sq)); AU.record(c->ev_join, s1); }
    if (sq != st) { STOCS_HIP_CHECK(hipStreamWaitEvent(st, c->ev_join, 0)); AU.wait(s0, c->ev_join); }   // the join needs both sides
    AU.use(s0, S->d_qkeys.p, false, "sorted Q keys", "join count"); AU.use(s0, S->d_qvals.p, false, "sorted Q pairs", "join count"); AU.use(s0, plan.jobs, false, "base jobs", "join count");
    if (dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[3], st));
    STOCS_TICK("gather+sort+records")
    // join: count pass + exclusive scan.  The quads themselves are produced on demand (materialise / resolve_picks_kernel)
    DevBuf<unsigned long long> d_qcnt;   // 64-bit: the total can exceed 2^32
    if ((rc = d_qcnt.alloc(totQ + 1)) || (rc = S->d_qoffe.alloc(totQ + 1))) return rc;
    hipLaunchKernelGGL(join_count_kernel<KeyT>, dim3((unsigned)((totQ + 255) / 256)), dim3(256), 0, st, S->args<KeyT>(c), d_qcnt.p);
    STOCS_HIP_CHECK(hipGetLastError());
    if (dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[4], st));
    size_t tmp_scan = 0;
    STOCS_HIP_CHECK(exclusive_scan(NULL, tmp_scan, d_qcnt.p, S->d_qoffe.p, totQ + 1, st));
    DevBuf<char> d_tmp_scan;
    if ((rc = d_tmp_scan.alloc(tmp_scan))) return rc;
    STOCS_HIP_CHECK(exclusive_scan(d_tmp_scan.p, tmp_scan, d_qcnt.p, S->d_qoffe.p, totQ + 1, st));
    // per-base offsets = scan value at the first Q entry of each base
    unsigned long long* qoff_at = (unsigned long long*)((char*)c->h_pin + PIN_VAR);   // pinned (sized by the caller)
    DevBuf<unsigned long long> d_boff;
    if ((rc = d_boff.alloc(nB + 1))) return rc;
    hipLaunchKernelGGL(base_offsets_kernel, dim3((unsigned)((nB + 1 + 255) / 256)), dim3(256), 0, st, S->d_qoffe.p, S->d_qoff.p, nB + 1, d_boff.p);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(hipMemcpyAsync(qoff_at, d_boff.p, 8 * (size_t)(nB + 1), hipMemcpyDeviceToHost, st));
    uint32_t* sort_err_pin = (uint32_t*)((char*)c->h_pin + PIN_CONGRUENT + 128);   // the own sort's error words (a look-back wait that ran into its bound)
    sort_err_pin[0] = sort_err_pin[1] = 0u;
    // (both sorts are joined into st by now)
    if (own_p) STOCS_HIP_CHECK(hipMemcpyAsync(&sort_err_pin[0], d_tmp.p + sort_own_err_offset(), 4, hipMemcpyDeviceToHost, st));
    if (own_q) STOCS_HIP_CHECK(hipMemcpyAsync(&sort_err_pin[1], d_tmp2.p + sort_own_err_offset(), 4, hipMemcpyDeviceToHost, st));
    if (dev_clock) STOCS_HIP_CHECK(hipEventRecord(c->ev_t[5], st));
    c->timing[0].lap(reduce ? "enqueue compact/sort/records/join/scan" : "enqueue gather/sort/records/join/scan");
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    AU.host_sync(s0);
    c->timing[0].lap("wait for the device (counts)");
    {   // the device's own account of that wait (every event has completed: the stream is idle, the Q side was joined into it)
        static const char* const what_all[5] = {"device: Q gather + sort (aux stream, from the fork)", "device: P gather + sort", "device: P records + wait for Q",
                                                "device: join count", "device: scan + offsets + read-back"};
        static const char* const what_red[6] = {"device: gathers + occupancy + survivor counts (both lists)", "device: Q sort (aux stream, from the fork; its compaction ran during the wait)",
                                                "device: P sort", "device: P records + wait for Q", "device: join count", "device: scan + offsets + read-back"};
        const int from[6] = {6, 0, 0, 2, 3, 4}, to[6] = {7, 1, 2, 3, 4, 5};
        static const char* const what_one[6] = {"device: gathers + occupancy + survivor counts (both lists in every launch)", "", "device: sort (P and Q as one list of 2 nB segments)",
                                                "device: P records", "device: join count", "device: scan + offsets + read-back"};
        for (int k = have_surv_clock ? 0 : 1; k < 6 && dev_clock; ++k) {
            if (one_stream && k == 1) continue;
            float ms = -1.0f;
            if (hipEventElapsedTime(&ms, c->ev_t[from[k]], c->ev_t[to[k]]) != hipSuccess) ms = -1.0f;
            CallTiming& T = c->timing[0];
            if (T.n < CallTiming::MAX_STEPS) { T.label[T.n] = one_stream ? what_one[k] : (have_surv_clock ? what_red[k] : what_all[k - 1]); T.ms[T.n] = (double)ms; ++T.n; }
        }
        c->timing[0].t_last = CallTiming::now_s();
    }
    STOCS_TICK("join count+scan")
    if (sort_err_pin[0] || sort_err_pin[1]) { set_error("stocs_find_congruent_all: the pair-list sort gave up waiting for a tile (sort32.hip)"); return STOCS_ERR_HIP; }
    for (int b = 0; b <= nB; ++b) c->quad_off[b] = qoff_at[b];
    return STOCS_OK;
}

}  // namespace stocs

using namespace stocs;

extern "C" {

void stocs_internal_free_congruent(stocs_ctx* c) {   // stocs_ctx_destroy: nothing is in flight any more
    if (c && c->cong) {
        CongruentState* S = (CongruentState*)c->cong;
        S->arena_state.destroy(); S->arena_tmp.destroy();
        if (S->h_stage) (void)hipHostFree(S->h_stage);
        if (S->d_plan) (void)hipFree(S->d_plan);
        if (S->d_trig) (void)hipFree(S->d_trig);
        delete S;
        c->cong = NULL;
    }
}

void stocs_internal_invalidate_congruent(stocs_ctx* c) {   // the counted state refers to bases of another scene
    if (c && c->cong) { CongruentState* S = (CongruentState*)c->cong; S->valid = false; S->hist_nB = 0; S->hist_P = S->hist_Q = 0; }
}

// Host evaluation of one cone query with both evaluations of cone_cells.h (no device needed): the direction-cell bitset
// from the reference's arithmetic alone, the one the kernels build (filter first, exact arithmetic where the filter
// cannot decide) and how many of the samples needed the exact path.  The two bitsets must be identical.
int stocs_cone_cells_host(const float* n3, float cos_alpha, uint32_t* exact_bits11, uint32_t* kernel_bits11, int* n_samples, int* n_undecided) {
    if (!n3 || !exact_bits11 || !kernel_bits11) return STOCS_ERR_INVALID;
    BaseJob J;
    memset(&J, 0, sizeof(J));
    J.cos_alpha = cos_alpha;
    fill_cone_table(&J);
    const float nepsilon = (float)((double)(1.0f / 7.0f) + 0.00001);
    const float half_inv_neps = (float)(0.5 / (double)nepsilon);
    for (int k = 0; k < 11; ++k) exact_bits11[k] = kernel_bits11[k] = 0;
    float q[4];
    quat_from_z(mk3(n3[0], n3[1], n3[2]), q);
    const ConeFilter cf = cone_filter_setup(q, J.cos_alpha, half_inv_neps);
    int undecided = 0;
    const float* trig = cone_trig() + (size_t)J.nb * STOCS_MAX_CONE * 2;
    for (int a = 0; a < J.nb; ++a) {
        const V3 d = mk3(J.sin_alpha * trig[2 * a], J.sin_alpha * trig[2 * a + 1], J.cos_alpha);
        const int ie = cone_cell_exact(q, d, nepsilon);
        if (ie >= 0) exact_bits11[ie >> 5] |= 1u << (ie & 31);
        int ik = cone_cell_filtered(cf, d.x, d.y);
        if (ik < 0) { undecided++; ik = ie; }
        if (ik >= 0) kernel_bits11[ik >> 5] |= 1u << (ik & 31);
    }
    if (n_samples) *n_samples = J.nb;
    if (n_undecided) *n_undecided = undecided;
    return STOCS_OK;
}

int stocs_find_congruent_all(stocs_ctx* c, int64_t* total_quads) { return stocs_internal_find_congruent(c, total_quads, 0, NULL); }

int stocs_internal_find_congruent(stocs_ctx* c, int64_t* total_quads, size_t max_bytes, int* too_big) {
    if (!c) return STOCS_ERR_INVALID;
    if (too_big) *too_big = 0;
    DeviceGuard dev_guard(c->device);
    const bool dbg = getenv("STOCS_DEBUG_TIMING") != NULL;
    double tprev = now_s();
    c->timing[0].begin();
    if (!c->index.built) { set_error("stocs_find_congruent_all: PPF index not built"); return STOCS_ERR_STATE; }
    const int nB = (int)c->bases.size();
    if (!c->cong) c->cong = new CongruentState();
    CongruentState* S = (CongruentState*)c->cong;
    S->valid = false;
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));   // the previous trial's buffers are about to be reused
    if (c->aux_stream) STOCS_HIP_CHECK(hipStreamSynchronize(c->aux_stream));   // (idle unless an earlier call failed half way)
    c->audit.on = getenv("STOCS_DEBUG_STREAMS") != NULL;
    c->audit.host_sync(0); if (c->aux_stream) c->audit.host_sync(1);
    c->audit.retire_all("stocs_find_congruent_all: the arena is recycled");
    c->timing[0].lap("entry synchronisation");
    { int rc0 = S->arena_state.reset(); if (rc0) return rc0; }
    tl_arena = &S->arena_state;
    if (!S->d_trig) {   // once per context (synchronous copy from static host memory)
        STOCS_HIP_CHECK(dev_malloc((void**)&S->d_trig, sizeof(float2) * (STOCS_MAX_CONE + 1) * STOCS_MAX_CONE));
        STOCS_HIP_CHECK(hipMemcpy(S->d_trig, cone_trig(), sizeof(float2) * (STOCS_MAX_CONE + 1) * STOCS_MAX_CONE, hipMemcpyHostToDevice));
    }
    // pinned block for everything this call reads back: plan totals, Q offsets (4 B per base), per-base quad offsets (8 B)
    { int rc0 = ensure_pinned(c, (size_t)PIN_VAR + 12 * ((size_t)nB + 1) + 256); if (rc0) return rc0; }
    c->timing[0].lap("arena reset + pinned block");
    if (dbg) { const double t_ = now_s(); fprintf(stderr, "[stocs congruent] %-18s %8.3f ms\n", "sync+reset", (t_ - tprev) * 1e3); tprev = t_; }
    c->quad_off.assign(nB + 1, 0);
    c->quad_id_bits = 16;
    if (total_quads) *total_quads = 0;
    if (nB == 0) return STOCS_OK;
    if (nB >= (1 << 20)) { set_error("too many bases"); return STOCS_ERR_INVALID; }
    const PpfIndex& ix = c->index;

    // ---- per base: the job record (invariants, cone table), then the plan of its two lookups ----
    std::vector<BaseJob> jobs(nB);
    const float eps_unit = c->prm.distance_threshold / c->ratio;  // getNormalizedEpsilon, pairCreationFunctor.h:141-143
    const int gridDepth = (int)(-log2f(eps_unit));                // normalset.h:117
    const int egSize = (int)pow(2.0, (double)gridDepth);          // :118
    const float cell = 1.f / egSize;                               // :119
    const float nepsilon = (float)((double)(1.0f / 7.0f) + 0.00001);  // normalset.h:86
    // ---- key layout ----
    const long long NC = (long long)egSize * egSize * egSize;
    // (8 bytes of run table + 2 of occupancy per (base, cell): up to 2.7 GB -- a trial batch brings thousands of bases, and 288 GB are there for it)
    const bool use_table = NC > 0 && NC * (long long)nB <= (long long)256 * 1024 * 1024;
    int base_bits = 1, id_bits = 1, cell_bits = 1;
    while ((1 << base_bits) < nB) base_bits++;
    while ((1 << id_bits) < c->nM) id_bits++;
    if (const char* e = getenv("STOCS_CONGRUENT_ID_BITS")) id_bits = std::max(id_bits, std::min(16, atoi(e)));   // keeps the wide-id form of the quad keys testable on small models
    {   // cells 0 .. limit-1 plus the all-ones "no cell" value
        const unsigned long long lim = use_table ? (unsigned long long)NC : ((unsigned long long)1 << 31);
        while (cell_bits < 40 && (((unsigned long long)1 << cell_bits) - 1ull) < lim) cell_bits++;
    }
    const bool wide = base_bits + cell_bits > 32 || getenv("STOCS_CONGRUENT_WIDE_KEYS") != NULL;   // env: keeps the 64-bit path testable
    // both pair lists are reduced to the entries with a partner cell when one byte per (base, cell) is a small table (count_pass)
    const bool reduce = use_table && ((unsigned long long)nB << cell_bits) <= (1ull << 29) && !getenv("STOCS_CONGRUENT_KEEP_ALL");
    // a batch's bases: the cone records wait until the device is busy (S->deferred, below)
    const bool defer_cone = nB >= 512 && nB <= PLAN_MAX_BASES && !getenv("STOCS_CONGRUENT_HOST_PLAN");
    S->deferred = nullptr;
    for (int b = 0; b < nB; ++b) {
        const BaseRec& B = c->bases[b];
        BaseJob& J = jobs[b];
        memset(&J, 0, sizeof(J));
        J.inv1 = B.inv1; J.inv2 = B.inv2;
        J.cell = cell; J.egSize = egSize;
        if (defer_cone) continue;
        J.cos_alpha = dot3(normalized3(c->h_spos[B.ids[1]] - c->h_spos[B.ids[0]]), normalized3(c->h_spos[B.ids[3]] - c->h_spos[B.ids[2]]));  // stocs.cpp:801-803
        fill_cone_table(&J);
    }
    // the planning buffer: jobs | base ids | error words | ranges | range counts | totals | P segments | Q segments | p_off | q_off |
    // segment offsets | result.  A lookup is at most 128 ranges, so every array has a bound that depends on nB alone.
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nb = (size_t)nB;
    const size_t o_jobs = 0, o_bids = o_jobs + al(sizeof(BaseJob) * nb), o_err = o_bids + al(16 * nb), o_rng = o_err + 256, o_nr = o_rng + al(2 * nb * 128 * 8),
                 o_tot = o_nr + al(2 * nb * 4), o_pseg = o_tot + al(2 * nb * 4), o_qseg = o_pseg + al(nb * 128 * sizeof(Segment)),
                 o_poff = o_qseg + al(nb * 128 * sizeof(Segment)), o_qoff = o_poff + al((nb + 1) * 4), o_spo = o_qoff + al((nb + 2) * 4), o_sqo = o_spo + al((nb + 1) * 4),
                 o_out = o_sqo + al((nb + 1) * 4), plan_bytes = o_out + 256, up_bytes = o_err + 256;
    if (S->plan_bytes < plan_bytes) {
        if (S->d_plan) (void)hipFree(S->d_plan);
        S->d_plan = NULL; S->plan_bytes = 0;
        STOCS_HIP_CHECK(dev_malloc((void**)&S->d_plan, 2 * plan_bytes));
        S->plan_bytes = 2 * plan_bytes;
    }
    const size_t host_plan_bytes = o_out;   // the host-planned form stages everything up to the result block
    if (S->stage_bytes < host_plan_bytes) {
        if (S->h_stage) (void)hipHostFree(S->h_stage);
        S->h_stage = NULL; S->stage_bytes = 0;
        STOCS_HIP_CHECK(pinned_malloc(&S->h_stage, 2 * host_plan_bytes));   // counted: a regrow inside a trial must show up
        S->stage_bytes = 2 * host_plan_bytes;
    }
    c->timing[0].lap("host: jobs + cone tables + buffers");
    char* h = (char*)S->h_stage;
    char* dpl = S->d_plan;
    PlanDev plan;
    plan.jobs = (BaseJob*)(dpl + o_jobs); plan.bids = (int32_t*)(dpl + o_bids); plan.err = (unsigned int*)(dpl + o_err);
    plan.psegs = (Segment*)(dpl + o_pseg); plan.qsegs = (Segment*)(dpl + o_qseg); plan.p_off = (uint32_t*)(dpl + o_poff); plan.q_off = (uint32_t*)(dpl + o_qoff);
    plan.n_pseg = 0; plan.n_qseg = 0;
    uint64_t totP = 0, totQ = 0;
    bool optimistic = false;
    const PlanOut* d_po = NULL; const PlanOut* po_pinned = NULL;
    std::vector<uint32_t>& q_off = S->h_qoff;
    q_off.assign(nB + 1, 0);
    {
        int32_t* bids = (int32_t*)(h + o_bids);
        for (int b = 0; b < nB; ++b) for (int k = 0; k < 4; ++k) bids[4 * b + k] = c->bases[b].ids[k];
        memset(h + o_err, 0, 256);
    }
    if (nB <= PLAN_MAX_BASES && !getenv("STOCS_CONGRUENT_HOST_PLAN")) {
        // on the device: 2 x nB small workgroups probe the bucket table, one workgroup lays the lists out; the host reads
        // back four totals and the Q offsets (it sizes the sorts with them) -- 0.2 ms of host work per trial otherwise
        memcpy(h + o_jobs, jobs.data(), sizeof(BaseJob) * nb);
        STOCS_HIP_CHECK(hipMemcpyAsync(dpl, h, up_bytes, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(plan_ranges_kernel, dim3((unsigned)nB, 2), dim3(64), 0, c->stream, ix.d_bucket_start, ix.tr, ix.rot, ix.NA, ix.nD, plan.bids,
                           (const float4*)c->d_spos, (const float4*)c->d_snrmw, nB, (uint2*)(dpl + o_rng), (uint32_t*)(dpl + o_nr), (uint32_t*)(dpl + o_tot));
        hipLaunchKernelGGL(plan_offsets_kernel, dim3(1), dim3(1024), 0, c->stream, nB, (const uint2*)(dpl + o_rng), (const uint32_t*)(dpl + o_nr), (const uint32_t*)(dpl + o_tot),
                           plan.jobs, plan.psegs, plan.qsegs, plan.p_off, plan.q_off, (uint32_t*)(dpl + o_spo), (uint32_t*)(dpl + o_sqo), (PlanOut*)(dpl + o_out), plan.err);
        hipLaunchKernelGGL(plan_segments_kernel, dim3((unsigned)((2 * nB + 3) / 4)), dim3(256), 0, c->stream, nB, (const uint2*)(dpl + o_rng), (const uint32_t*)(dpl + o_nr),
                           (const uint32_t*)(dpl + o_tot), plan.psegs, plan.qsegs, (const uint32_t*)plan.p_off, (const uint32_t*)plan.q_off, (const uint32_t*)(dpl + o_spo),
                           (const uint32_t*)(dpl + o_sqo));
        STOCS_HIP_CHECK(hipGetLastError());
        // read-backs land in the pinned block (a copy into pageable memory -- a stack variable, a std::vector -- takes the
        // runtime's staging path): totals in the fixed slot, the Q offsets behind the per-base quad offsets of count_pass
        static_assert(sizeof(PlanOut) <= 256, "PlanOut must fit its pinned slot");
        PlanOut* po_pin = (PlanOut*)((char*)c->h_pin + PIN_CONGRUENT);
        uint32_t* qoff_pin = (uint32_t*)((char*)c->h_pin + PIN_VAR + 8 * (nb + 1));
        STOCS_HIP_CHECK(hipMemcpyAsync(po_pin, dpl + o_out, sizeof(PlanOut), hipMemcpyDeviceToHost, c->stream));
        if (!reduce) STOCS_HIP_CHECK(hipMemcpyAsync(qoff_pin, plan.q_off, 4 * (nb + 1), hipMemcpyDeviceToHost, c->stream));   // (reduced lists: their own offsets come later)
        c->timing[0].lap("enqueue plan upload + kernels + read-back");
        if (defer_cone) {
            // staged in the part of the pinned block that only the host-planned form uses (ranges), uploaded and patched into the device's
            // jobs on the context's stream -- behind the plan kernels, ahead of the join
            float4* hc = (float4*)(h + o_rng);
            BaseJob* dj = plan.jobs;
            S->deferred = [c, hc, dj, nB]() -> int {
                for (int b = 0; b < nB; ++b) {
                    const BaseRec& B = c->bases[b];
                    BaseJob J;
                    J.cos_alpha = dot3(normalized3(c->h_spos[B.ids[1]] - c->h_spos[B.ids[0]]), normalized3(c->h_spos[B.ids[3]] - c->h_spos[B.ids[2]]));  // stocs.cpp:801-803
                    fill_cone_table(&J);
                    float nb_bits; memcpy(&nb_bits, &J.nb, 4);
                    hc[b] = make_float4(J.cos_alpha, J.sin_alpha, nb_bits, 0.f);
                }
                DevBuf<float4> d_cone;
                int rc1 = d_cone.alloc((size_t)nB);
                if (rc1) return rc1;
                STOCS_HIP_CHECK(hipMemcpyAsync(d_cone.p, hc, sizeof(float4) * (size_t)nB, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(patch_cone_kernel, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, c->stream, dj, (const float4*)d_cone.p, nB);
                STOCS_HIP_CHECK(hipGetLastError());
                return STOCS_OK;
            };
        }
        // ONE sizing synchronisation point instead of two (plan totals, then survivors' totals): when an earlier trial of this scene
        // has shown how long the lists get, buffers and launches are sized by a capacity (1.6 x that, per base), the kernels read the
        // planned totals from the device, and the plan's totals come back together with the survivors' (count_pass).  A plan beyond
        // the capacity is detected there and redone with exact sizes.  Not for a trial batch under a memory ceiling (its caller
        // needs the totals first) and not for the first trial of a scene.
        optimistic = reduce && S->hist_nB > 0 && !getenv("STOCS_CONGRUENT_EXACT_SIZES");
        if (optimistic) {
            const char* ce = getenv("STOCS_CONGRUENT_CAPACITY");     // (tests: a factor below 1 forces the redo with exact sizes)
            const double scale = (ce ? atof(ce) : 1.6) * (double)nB / (double)S->hist_nB, slack = ce ? 1.0 : 1048576.0;
            totP = (uint64_t)std::min(4.0e9, (double)S->hist_P * scale + slack);
            totQ = (uint64_t)std::min(4.0e9, (double)S->hist_Q * scale + slack);
            // a piece of a trial batch (too_big: its caller wants to hear when the lists do not fit the ceiling) goes this way only when the
            // capacities are far below the ceiling -- small frames, where the plan's round trip is a tenth of the piece; near the ceiling the
            // totals are read first, as in round 4
            if (too_big && max_bytes && (double)(totP + totQ) * 64.0 + (use_table ? (double)(NC * nB) * 8.0 : 0.0) + (double)((size_t)nB << cell_bits) / 4.0 > 0.25 * (double)max_bytes) {
                optimistic = false; totP = 0; totQ = 0;
            }
        }
        if (optimistic) {
            d_po = (const PlanOut*)(dpl + o_out); po_pinned = po_pin;
        } else {
            STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
            c->timing[0].lap("wait for the device (plan)");
            const PlanOut po = *po_pin;
            if (!reduce) memcpy(q_off.data(), qoff_pin, 4 * (nb + 1));
            if (po.overflow) { if (too_big) { *too_big = 1; return STOCS_OK; } set_error("pair lists exceed 2^32 entries"); return STOCS_ERR_CAPACITY; }
            totP = po.totP; totQ = po.totQ; plan.n_pseg = (int)po.n_pseg; plan.n_qseg = (int)po.n_qseg;
            S->hist_P = po.totP; S->hist_Q = po.totQ; S->hist_nB = nB;
        }
        STOCS_TICK("plan (device)")
    } else {
        // on the host (kept for very many bases and for A/B): the same ranges from the host copy of the bucket table
        std::vector<Segment> psegs, qsegs;
        std::vector<std::pair<uint32_t, uint32_t> > pr, qr;
        std::vector<int> keys8((size_t)nB * 8);
        for (int b = 0; b < nB; ++b) {   // the keys first, touching the bucket table ahead of the planning loop
            const BaseRec& B = c->bases[b];
            int* K1 = &keys8[(size_t)b * 8];
            int* K2 = K1 + 4;
            ppf_compute(c->h_spos[B.ids[0]], c->h_snrm[B.ids[0]], c->h_spos[B.ids[1]], c->h_snrm[B.ids[1]], ix.tr, ix.rot, K1);  // stocs.cpp:771
            ppf_compute(c->h_spos[B.ids[2]], c->h_snrm[B.ids[2]], c->h_spos[B.ids[3]], c->h_snrm[B.ids[3]], ix.tr, ix.rot, K2);  // stocs.cpp:772
            prefetch_lookup(ix, K1);
            prefetch_lookup(ix, K2);
        }
        for (int b = 0; b < nB; ++b) {
            BaseJob& J = jobs[b];
            const int* K1 = &keys8[(size_t)b * 8];
            const int* K2 = K1 + 4;
            plan_lookup(ix, K1, &pr);
            plan_lookup(ix, K2, &qr);
            uint64_t np = 0, nq = 0;
            for (size_t r = 0; r < pr.size(); ++r) np += pr[r].second - pr[r].first;
            for (size_t r = 0; r < qr.size(); ++r) nq += qr[r].second - qr[r].first;
            if (np == 0 || nq == 0) { np = 0; nq = 0; pr.clear(); qr.clear(); }  // stocs.cpp:788
            J.p_off = (uint32_t)totP; J.p_len = (uint32_t)np; J.q_off = (uint32_t)totQ; J.q_len = (uint32_t)nq;
            uint32_t d = (uint32_t)totP;
            for (size_t r = 0; r < pr.size(); ++r) { Segment sg = {pr[r].first, pr[r].second - pr[r].first, d, (uint32_t)b}; psegs.push_back(sg); d += sg.len; }
            d = (uint32_t)totQ;
            for (size_t r = 0; r < qr.size(); ++r) { Segment sg = {qr[r].first, qr[r].second - qr[r].first, d, (uint32_t)b}; qsegs.push_back(sg); d += sg.len; }
            q_off[b] = (uint32_t)totQ;
            totP += np; totQ += nq;
            if (totP >= 0xFFFF0000ull || totQ >= 0xFFFF0000ull) { if (too_big) { *too_big = 1; return STOCS_OK; } set_error("pair lists exceed 2^32 entries"); return STOCS_ERR_CAPACITY; }
        }
        q_off[nB] = (uint32_t)totQ;
        if (psegs.size() > nb * 128 || qsegs.size() > nb * 128) { set_error("internal: more than 128 ranges per lookup"); return STOCS_ERR_STATE; }
        memcpy(h + o_jobs, jobs.data(), sizeof(BaseJob) * nb);
        memcpy(h + o_pseg, psegs.data(), sizeof(Segment) * psegs.size());
        memcpy(h + o_qseg, qsegs.data(), sizeof(Segment) * qsegs.size());
        memcpy(h + o_qoff, q_off.data(), 4 * (nb + 1));
        STOCS_HIP_CHECK(hipMemcpyAsync(dpl, h, up_bytes, hipMemcpyHostToDevice, c->stream));
        STOCS_HIP_CHECK(hipMemcpyAsync(dpl + o_pseg, h + o_pseg, sizeof(Segment) * psegs.size(), hipMemcpyHostToDevice, c->stream));
        STOCS_HIP_CHECK(hipMemcpyAsync(dpl + o_qseg, h + o_qseg, sizeof(Segment) * qsegs.size(), hipMemcpyHostToDevice, c->stream));
        STOCS_HIP_CHECK(hipMemcpyAsync(dpl + o_qoff, h + o_qoff, 4 * (nb + 1), hipMemcpyHostToDevice, c->stream));
        plan.n_pseg = (int)psegs.size(); plan.n_qseg = (int)qsegs.size();
        c->timing[0].lap("plan on the host + upload");
        STOCS_TICK("plan (host)")
    }
    if (dbg) fprintf(stderr, "[stocs congruent] %s totP %llu totQ %llu segs %d %d\n", optimistic ? "capacities" : "planned", (unsigned long long)totP, (unsigned long long)totQ, plan.n_pseg, plan.n_qseg);
    if (totP == 0 || totQ == 0) return STOCS_OK;
    if (id_bits > 16) { set_error("|M| = %d: model ids beyond 16 bits do not fit the packed pairs and quads", c->nM); return STOCS_ERR_CAPACITY; }

    auto reserve_arena = [&](uint64_t nP, uint64_t nQ, bool* over) -> int {
        // everything this call allocates, estimated up front: one slab, one hipMalloc in a context's lifetime (if sizes stay put)
        const size_t kb = wide ? 8 : 4;
        const size_t tables = use_table ? (size_t)(NC * nB) * 8 : 0;
        const size_t per_entry = 3 * kb + 8 + 16 + 8 + (reduce ? kb + 4 : 0);   // (the compacted copies of the reduced form)
        const size_t occ = reduce ? 2 * (((size_t)nB << cell_bits) / 8 + 8) : 0;
        const size_t need = (size_t)nP * per_entry + (size_t)nQ * per_entry + tables + occ + ((size_t)48 << 20);
        if (max_bytes && need > max_bytes && nB > 1) { *over = true; return STOCS_OK; }
        // (under a ceiling -- a piece of a trial batch -- the slab is 1.5 x the need, not twice it: 40 Cm trials need 36 GB)
        const int rc_r = S->arena_state.reserve(need, (max_bytes && need > ((size_t)4 << 30)) ? 1.5 : 2.0);   // (small pieces: room for the next call's capacities, which follow THIS call's totals)
        if (rc_r == STOCS_ERR_NOMEM && max_bytes && nB > 1) { *over = true; return STOCS_OK; }   // the device is fuller than the ceiling assumes: the caller halves the piece
        return rc_r;
    };
    {
        bool over = false;
        int rc0 = reserve_arena(totP, totQ, &over);
        if (rc0) return rc0;
        if (over) { if (too_big) *too_big = 1; return STOCS_OK; }   // the caller splits its base set
    }
    c->timing[0].lap("arena reserve");
    S->nB = nB; S->totP = (uint32_t)totP; S->totQ = (uint32_t)totQ; S->nepsilon = nepsilon;
    S->half_inv_neps = (float)(0.5 / (double)nepsilon);
    S->NC = NC; S->use_table = use_table; S->wide = wide; S->reduce = reduce;
    {   // two points of one position cell are at most a cell diagonal apart (cell edge = ratio / egSize < 2 epsilon); the
        // float roundings of the two world-space points and of the unit-cube coordinates are far below the 1e-5 m allowed for
        const double cell_world = (double)c->ratio / (double)egSize, diag2 = 3.0 * (cell_world * 1.001 + 1e-5) * (cell_world * 1.001 + 1e-5);
        S->close_cells = diag2 < 0.999 * (double)c->prm.distance_threshold && !getenv("STOCS_CONGRUENT_DISTANCE_GATE");
    }
    S->id_bits = id_bits; S->base_bits = base_bits; S->cell_bits = cell_bits;
    S->base_in_key = 4 * id_bits + base_bits <= 64;
    S->no_quads = false;
    int rc = wide ? count_pass<uint64_t>(c, S, plan, dbg, tprev, d_po, po_pinned) : count_pass<uint32_t>(c, S, plan, dbg, tprev, d_po, po_pinned);
    if (optimistic && (rc == STOCS_OK || rc == 1)) {   // (the read-back of count_pass brought the plan's totals; on an error the copy may not have landed: the history stays as it was)
        const PlanOut po = *po_pinned;
        if (po.overflow) { if (too_big) { *too_big = 1; return STOCS_OK; } set_error("pair lists exceed 2^32 entries"); return STOCS_ERR_CAPACITY; }
        S->hist_P = po.totP; S->hist_Q = po.totQ; S->hist_nB = nB;
        if (rc == 1) {
            // the plan outgrew the capacities: once more with the sizes now known.  The base jobs carry the survivors' offsets by now:
            // the layout kernel writes the planned ones again (its inputs are still in the planning buffer)
            c->timing[0].lap("plan beyond the capacities: redone with exact sizes");
            const size_t nb2 = (size_t)nB;
            char* dpl2 = S->d_plan;
            auto al2 = [](size_t x) { return (x + 255) & ~(size_t)255; };
            const size_t o_bids2 = al2(sizeof(BaseJob) * nb2), o_err2 = o_bids2 + al2(16 * nb2), o_rng2 = o_err2 + 256, o_nr2 = o_rng2 + al2(2 * nb2 * 128 * 8),
                         o_tot2 = o_nr2 + al2(2 * nb2 * 4), o_pseg2 = o_tot2 + al2(2 * nb2 * 4), o_qseg2 = o_pseg2 + al2(nb2 * 128 * sizeof(Segment)),
                         o_poff2 = o_qseg2 + al2(nb2 * 128 * sizeof(Segment)), o_qoff2 = o_poff2 + al2((nb2 + 1) * 4), o_spo2 = o_qoff2 + al2((nb2 + 2) * 4),
                         o_sqo2 = o_spo2 + al2((nb2 + 1) * 4), o_out2 = o_sqo2 + al2((nb2 + 1) * 4);
            hipLaunchKernelGGL(plan_offsets_kernel, dim3(1), dim3(1024), 0, c->stream, nB, (const uint2*)(dpl2 + o_rng2), (const uint32_t*)(dpl2 + o_nr2), (const uint32_t*)(dpl2 + o_tot2),
                               plan.jobs, plan.psegs, plan.qsegs, plan.p_off, plan.q_off, (uint32_t*)(dpl2 + o_spo2), (uint32_t*)(dpl2 + o_sqo2), (PlanOut*)(dpl2 + o_out2), plan.err);
            STOCS_HIP_CHECK(hipGetLastError());
            { int rc0 = S->arena_state.reset(); if (rc0) return rc0; }
            tl_arena = &S->arena_state;
            bool over = false;
            int rc0 = reserve_arena(po.totP, po.totQ, &over);
            if (rc0) return rc0;
            if (over) { if (too_big) *too_big = 1; return STOCS_OK; }   // (a batch's piece: the caller splits its base set)
            plan.n_pseg = (int)po.n_pseg; plan.n_qseg = (int)po.n_qseg;
            S->totP = (uint32_t)po.totP; S->totQ = (uint32_t)po.totQ;
            S->no_quads = false;
            if (po.totP == 0 || po.totQ == 0) return STOCS_OK;
            rc = wide ? count_pass<uint64_t>(c, S, plan, dbg, tprev) : count_pass<uint32_t>(c, S, plan, dbg, tprev);
        }
    }
    if (rc) return rc;
    if (!c->audit.violations.empty()) {   // STOCS_DEBUG_STREAMS: a use without an event edge between the two streams
        set_error("stocs_find_congruent_all: %zu stream-ordering violation(s); first: %s", c->audit.violations.size(), c->audit.violations[0].c_str());
        c->audit.violations.clear();
        return STOCS_ERR_STATE;
    }
    if (S->no_quads) return STOCS_OK;   // as with empty lists: nothing to materialise, every base has zero quads
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
    S->small_ready = false;
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

// quads of base `slot` at the given ranks of its WALK order (by position cell of the Q pair, then index position of the
// Q pair, then index position of the P pair; see the head of this file) -- what stocs_make_transforms samples from when
// a base has >= max quads
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
    S->small_ready = false;
    DevBuf<Pick> d_picks; DevBuf<uint64_t> d_keys;
    int rc;
    if ((rc = d_picks.alloc(n)) || (rc = d_keys.alloc(n))) return rc;
    if ((rc = ensure_pinned(c, (size_t)PIN_VAR + 8 * (size_t)n + 256))) return rc;
    uint64_t* keys = (uint64_t*)((char*)c->h_pin + PIN_VAR);
    unsigned int* n_err_pin = (unsigned int*)((char*)c->h_pin + PIN_CONGRUENT);
    STOCS_HIP_CHECK(hipMemcpyAsync(d_picks.p, picks.data(), sizeof(Pick) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (S->wide)
        hipLaunchKernelGGL(resolve_picks_kernel<uint64_t>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, S->args<uint64_t>(c), S->d_qoffe.p, d_picks.p, n,
                           (const uint64_t*)NULL, (const unsigned long long*)NULL, S->d_bids.p, (XformJobC*)NULL, d_keys.p, S->d_err.p);
    else
        hipLaunchKernelGGL(resolve_picks_kernel<uint32_t>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, S->args<uint32_t>(c), S->d_qoffe.p, d_picks.p, n,
                           (const uint64_t*)NULL, (const unsigned long long*)NULL, S->d_bids.p, (XformJobC*)NULL, d_keys.p, S->d_err.p);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(hipMemcpyAsync(keys, d_keys.p, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(n_err_pin, S->d_err.p, 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    const unsigned int n_err = *n_err_pin;
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
// *d_unresolved_out: device counter of the picks the kernel could not resolve (0 while counts and join agree); the
// caller reads it behind its own synchronisation point, this call does not wait for the device
// First half of stocs_make_transforms' device work, launched BEFORE the host draws its picks: the bases with fewer quads than
// the per-base maximum are materialised and sorted (they are used whole, in the std::set order).  Which bases those are
// follows from the counts alone, so the fill + sort run while the host is busy with the seeded subsets of the large bases.
int stocs_internal_prepare_small(stocs_ctx* c, int max_per_base) {
    CongruentState* S = (CongruentState*)c->cong;
    if (!S || !S->valid) { set_error("stocs_make_transforms: no congruent state (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    { int rc0 = S->arena_tmp.reset(); if (rc0) return rc0; }
    tl_arena = &S->arena_tmp;
    S->small_ready = true; S->small_any = false;
    std::vector<char> sel(S->nB, 0);
    for (int b = 0; b < S->nB; ++b) {
        const unsigned long long nq = c->quad_off[b + 1] - c->quad_off[b];
        if (nq > 0 && nq < (unsigned long long)max_per_base) { sel[b] = 1; S->small_any = true; }
    }
    if (!S->small_any) return STOCS_OK;
    std::vector<unsigned long long>& off = S->h_off;
    int rc;
    if ((rc = materialise(c, S, sel, &S->d_small_sorted, &off)) || (rc = S->d_small_soff.alloc(off.size()))) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(S->d_small_soff.p, off.data(), 8 * off.size(), hipMemcpyHostToDevice, c->stream));
    return STOCS_OK;
}

// picks4_dev != NULL: the picks are on the device already (drawn there; stocs_internal_prepare_small has run)
int stocs_internal_make_jobs(stocs_ctx* c, const int32_t* picks4_host, const int32_t* picks4_dev, int n, void* d_jobs_out, const unsigned int** d_unresolved_out) {
    if (d_unresolved_out) *d_unresolved_out = NULL;
    CongruentState* S = (CongruentState*)c->cong;
    const bool prepared = S && S->small_ready;
    if (S) S->small_ready = false;
    if (n <= 0) return STOCS_OK;
    if (!S || !S->valid) { set_error("stocs_make_transforms: no congruent state (call stocs_find_congruent_all first)"); return STOCS_ERR_STATE; }
    const Pick* picks = (const Pick*)picks4_host;
    const uint64_t* d_sorted = NULL; const unsigned long long* d_soff = NULL;
    DevBuf<uint64_t> d_sorted_here; DevBuf<unsigned long long> d_soff_here; DevBuf<Pick> d_picks;
    int rc;
    if (picks4_dev && !prepared) { set_error("internal: device picks without the small bases prepared"); return STOCS_ERR_STATE; }
    if (prepared) {   // the temporaries of stocs_internal_prepare_small are still in the arena
        tl_arena = &S->arena_tmp;
        if (S->small_any) { d_sorted = S->d_small_sorted.p; d_soff = S->d_small_soff.p; }
    } else {
        { int rc0 = S->arena_tmp.reset(); if (rc0) return rc0; }
        tl_arena = &S->arena_tmp;
        std::vector<char> sel(S->nB, 0);
        bool any_sorted = false;
        for (int i = 0; i < n; ++i) if (picks[i].sorted) { sel[picks[i].base] = 1; any_sorted = true; }
        std::vector<unsigned long long>& off = S->h_off;
        if (any_sorted) {
            if ((rc = materialise(c, S, sel, &d_sorted_here, &off)) || (rc = d_soff_here.alloc(off.size()))) return rc;
            STOCS_HIP_CHECK(hipMemcpyAsync(d_soff_here.p, off.data(), 8 * off.size(), hipMemcpyHostToDevice, c->stream));
            d_sorted = d_sorted_here.p; d_soff = d_soff_here.p;
        }
    }
    if (picks4_dev) {
        d_picks.p = (Pick*)picks4_dev;
    } else {
        if ((rc = d_picks.alloc(n))) return rc;
        STOCS_HIP_CHECK(hipMemcpyAsync(d_picks.p, picks, sizeof(Pick) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    }
    if (S->wide)
        hipLaunchKernelGGL(resolve_picks_kernel<uint64_t>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, S->args<uint64_t>(c), S->d_qoffe.p, d_picks.p, n, d_sorted,
                           d_soff, S->d_bids.p, (XformJobC*)d_jobs_out, (uint64_t*)NULL, S->d_err.p);
    else
        hipLaunchKernelGGL(resolve_picks_kernel<uint32_t>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, S->args<uint32_t>(c), S->d_qoffe.p, d_picks.p, n, d_sorted,
                           d_soff, S->d_bids.p, (XformJobC*)d_jobs_out, (uint64_t*)NULL, S->d_err.p);
    STOCS_HIP_CHECK(hipGetLastError());
    if (d_unresolved_out) *d_unresolved_out = S->d_err.p;
    return STOCS_OK;
}

}  // extern "C"
