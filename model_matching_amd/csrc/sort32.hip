// sort32.hip -- device-wide STABLE radix sort of (u32 key, u32 value) pairs, hand-written for gfx950 and specialised to the pair lists of
// the congruent-set phase.  It stands where the reference keeps, per base, a pointer grid of std::vector<int> per position cell and
// inserts one pair at a time (IndexedNormalSet::addElement, reference include/super4pcs/accelerators/normalset.hpp:114-131; the loops of
// find_congruent_sets_on_model, src/stocs.cpp:806-866): a run of equal keys IS a cell's vector in insertion order -- hence stable.
//
// What the key is: (base << cell_bits | position cell), and the list comes out of the gather BASE-MAJOR -- all entries of base 0, then
// base 1, ... -- so the base bits are sorted already and only the cell bits (15-16) have to be.  Rounds 2-4 called rocPRIM's
// radix_sort_pairs over all significant bits (22 of a single trial's Q list: three 8-bit passes; 28 of a 40-trial piece: four).  This
// sort is SEGMENTED by base: tiles never straddle a base, the digit histogram and the scatter destinations are per (base, digit), and
// two passes (8 + 7 bits of the cell) sort any number of bases.  Per pass ONE launch (onesweep):
//   * a tile of 4 096 - 16 384 pairs per workgroup, wave-striped loads (slot j of lane l: the order of memory = (wavefront, slot, lane));
//   * the tile's digit counts first (plain LDS adds) and PUBLISHED at once, so the tiles behind find them ready;
//   * ranks inside a wavefront: the lanes that share a digit in a slot find each other through one 64-bit word per digit in LDS (every lane
//     ORs its bit in, reads the word back, the set's first lane adds the set to the digit's counter and clears the word) -- 3 LDS round trips
//     and ~25 instructions per key where one ballot per digit bit took ~90; a slot whose 64 lanes all hold one digit skips LDS;
//   * decoupled look-back per digit over the EARLIER TILES OF THE SAME BASE: one packed word per (tile, digit) -- 2 state bits + 30 value
//     bits, relaxed agent-scope atomics; the word carries its own value, so no fence (an agent-scope release would write the L2 back);
//     four predecessors are requested at a time;
//   * pairs reordered through LDS so that the scatter writes runs of consecutive addresses;
//   * tiles take their index from a ticket, so a tile only ever waits for tiles that already run, and every wait is bounded in wall-clock
//     time and reports instead of hanging.
// Bases of at most 2 048 pairs get no tiles: seg_small_sort_kernel sorts each whole in LDS, all passes in one launch.
// No fill launch of its own: control block and histograms are zeroed by the tile-table kernel, the look-back words of pass 0 by the histogram
// kernel, those of pass p + 1 by the tiles of pass p.
// HBM-bound byte work (16 B read + written per pair and pass); no MFMA.
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

#define SORT_MAX_PASS 4

struct SortCtl {                      // zero-filled once per sort
    uint32_t ticket[SORT_MAX_PASS];
    uint32_t err;                     // a look-back wait ran into its bound (never seen; the caller reports it)
    uint32_t n_tiles;                 // written by seg_tiles_kernel
    uint32_t pad[2];
};

struct SortPlan {
    int n_pass;
    int shift[SORT_MAX_PASS], bits[SORT_MAX_PASS];
};

static SortPlan sort_plan(unsigned b0, unsigned b1) {
    SortPlan P;
    const int total = (int)b1 - (int)b0;
    P.n_pass = std::max(1, (total + 7) / 8);
    int s = (int)b0, left = total;
    for (int p = 0; p < P.n_pass; ++p) {            // balanced digits, the wider ones first
        const int w = (left + (P.n_pass - p) - 1) / (P.n_pass - p);
        P.shift[p] = s; P.bits[p] = std::max(w, 1);
        s += w; left -= w;
    }
    for (int p = P.n_pass; p < SORT_MAX_PASS; ++p) { P.shift[p] = 0; P.bits[p] = 0; }
    return P;
}

// Segments -> tiles.  seg_off[0 .. n_seg]: where every segment (base) begins in the list; seg_off == NULL: one segment [0, n).  A segment of
// L pairs is ceil(L / tile) tiles; tile_first = exclusive scan, tile_seg[tile] = its segment.  One workgroup (the table is tiny next to the list).
__global__ __launch_bounds__(1024) void seg_tiles_kernel(const uint32_t* __restrict__ seg_off, uint32_t n_seg, uint32_t n, uint32_t tile, uint32_t max_tiles,
                                                         uint32_t* __restrict__ tile_first, uint32_t* __restrict__ tile_seg, uint32_t* __restrict__ own_off, SortCtl* __restrict__ ctl,
                                                         uint32_t* __restrict__ hist, uint32_t hist_words, uint32_t small_cap) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    if (blockIdx.x > 0) {                                         // workgroups 1..: the sort's one zero fill -- the digit histograms
        const uint32_t i0 = ((blockIdx.x - 1u) * 1024u + t) * 4u;
        if (i0 + 3u < hist_words) *(uint4*)(hist + i0) = make_uint4(0u, 0u, 0u, 0u);
        else for (uint32_t i = i0; i < hist_words; ++i) hist[i] = 0u;
        return;
    }
    if (t < SORT_MAX_PASS) ctl->ticket[t] = 0u;                    // workgroup 0: the control block (n_tiles is written below), then the tile table
    if (t == SORT_MAX_PASS) ctl->err = 0u;
    if (t == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_seg; b0 += 1024u) {
        const uint32_t b = b0 + t;
        uint32_t lo = 0, hi = 0;
        if (b < n_seg) { lo = seg_off ? seg_off[b] : 0u; hi = seg_off ? seg_off[b + 1] : n; lo = min(lo, n); hi = min(max(hi, lo), n); }
        const uint32_t nt = (hi - lo) > small_cap ? (hi - lo + tile - 1u) / tile : 0u;     // (short segments are sorted whole by seg_small_sort_kernel)
        uint32_t inc = nt;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += a; }
        if (lane == 63u) s_w[w] = inc;
        __syncthreads();
        uint32_t pre = s_carry;
        for (uint32_t x = 0; x < w; ++x) pre += s_w[x];
        const uint32_t first = pre + inc - nt;
        if (b < n_seg) {
            tile_first[b] = first;
            own_off[b] = lo;                                    // (a private copy: the caller's offsets may be rewritten while later passes run)
            if (b + 1 == n_seg) { own_off[n_seg] = hi; tile_first[n_seg] = first + nt; ctl->n_tiles = min(first + nt, max_tiles); }
            for (uint32_t k = 0; k < nt && first + k < max_tiles; ++k) tile_seg[first + k] = b;
        }
        __syncthreads();
        if (t == 1023u) s_carry = pre + inc;
        __syncthreads();
    }
}

// Adds `1` for every lane's digit d into cnt[d] (LDS) without serialising on the long runs of one cell that the pair lists are made of
// (an entry whose invariant is 0 sits at its pair's first point, and the index lists the ~150 pairs of a point together): a wavefront
// whose lanes all hold one digit adds once; any other goes through plain LDS atomics.
__device__ __forceinline__ void wave_count_digit(uint32_t* cnt, uint32_t d, bool valid) {
    const unsigned long long vm = __ballot(valid);
    if (!vm) return;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)__builtin_ctzll(vm));
    if (__ballot(valid && d != d0) == 0ull) {                    // one digit in every lane: one add
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(vm)) atomicAdd(&cnt[d0], (uint32_t)__popcll(vm));
        return;
    }
    if (valid) atomicAdd(&cnt[d], 1u);
}

// digit histograms per (pass, segment) in one read of the keys; also clears the look-back words of pass 0
// SUB > 1 (lists of a single trial: a few hundred tiles on 256 CUs): SUB workgroups share a tile of the passes, each counting a SUB-th of it with
// KPT / SUB loads per thread -- the kernel is a chain of three round trips (keys, LDS counters, device counters) and had one workgroup per tile
// walk it with 16 loads per thread (30 us for the 3.5 M pairs of a Cm trial: 0.5 TB/s).
template <int NW, int KPT_ALL, int SUB>
__global__ __launch_bounds__(64 * NW) void seg_hist_kernel(const uint32_t* __restrict__ keys, SortPlan P, const SortCtl* __restrict__ ctl, const uint32_t* __restrict__ tile_first,
                                                           const uint32_t* __restrict__ tile_seg, const uint32_t* __restrict__ seg_off, uint32_t n_seg,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ lb0) {
    constexpr uint32_t THREADS = 64u * NW, TILE = THREADS * KPT_ALL;
    constexpr int KPT = KPT_ALL / SUB;
    static_assert(KPT * SUB == KPT_ALL, "the parts of a tile are equal");
    __shared__ uint32_t s_h[SORT_MAX_PASS][256];
    const uint32_t tile = blockIdx.x / SUB, part = blockIdx.x % SUB, t = threadIdx.x;
    if (tile >= ctl->n_tiles) return;
    for (uint32_t i = t; i < SORT_MAX_PASS * 256u; i += THREADS) (&s_h[0][0])[i] = 0u;
    if (t < 256u && part == 0u) lb0[(size_t)tile * 256u + t] = 0u;
    __syncthreads();
    const uint32_t b = tile_seg[tile];
    const uint32_t tlo = seg_off[b] + (tile - tile_first[b]) * TILE, hi = min(seg_off[b + 1], tlo + TILE);
    const uint32_t lo = tlo + part * (TILE / SUB);                // (a part beyond the tile's end counts nothing: every i >= hi)
    uint32_t k[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {                               // every load of the tile in flight before the first count
        const uint32_t i = lo + (uint32_t)j * THREADS + t;
        k[j] = i < hi ? keys[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const bool v = lo + (uint32_t)j * THREADS + t < hi;
        for (int p = 0; p < P.n_pass; ++p) wave_count_digit(&s_h[p][0], (k[j] >> P.shift[p]) & ((1u << P.bits[p]) - 1u), v);
    }
    __syncthreads();
    if (t < 256u)
        for (int p = 0; p < P.n_pass; ++p) {
            const uint32_t c = s_h[p][t];
            if (c) atomicAdd(&hist[((size_t)p * n_seg + b) * 256u + t], c);
        }
}

#define LB_PARTIAL 0x40000000u
#define LB_COMPLETE 0x80000000u
#define LB_VALUE 0x3FFFFFFFu

template <int NW, int KPT>
__global__ __launch_bounds__(64 * NW) void seg_onesweep_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout,
                                                               uint32_t* __restrict__ vout, int pass, int shift, int bits, SortCtl* __restrict__ ctl,
                                                               const uint32_t* __restrict__ tile_first, const uint32_t* __restrict__ tile_seg, const uint32_t* __restrict__ seg_off,
                                                               uint32_t n_seg, const uint32_t* __restrict__ hist, uint32_t* __restrict__ lb, uint32_t* __restrict__ lb_next) {
    constexpr uint32_t THREADS = 64u * NW, TILE = THREADS * KPT;
    __shared__ uint32_t s_wh[NW][256];        // per wavefront: digit counters, then the digit's first slot for that wavefront
    __shared__ uint32_t s_gpos[256];          // digit -> (global position of the tile's first key of that digit) - (its slot in the tile)
    // (the staging area of the scatter doubles as the wavefronts' digit -> lane-set table of the ranking step: 2 KB per wavefront)
    __shared__ uint32_t s_kv[2 * TILE];
    uint32_t* const s_k = s_kv; uint32_t* const s_v = s_kv + TILE;
    unsigned long long* const s_m = (unsigned long long*)s_kv + (size_t)(threadIdx.x >> 6) * 256;
    static_assert(2 * TILE * 4 >= NW * 256 * 8, "the lane-set tables fit the staging area");
    __shared__ uint32_t s_part[2][4];
    __shared__ uint32_t s_tile;
    const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63u;
    const bool own = t < 256u;                                    // threads 0..255 each own one digit
    const uint32_t dmask = (1u << bits) - 1u;
    if (t == 0) s_tile = atomicAdd(&ctl->ticket[pass], 1u);      // tiles start in ticket order: a tile waits only for tiles that run
    for (uint32_t i = t; i < (uint32_t)NW * 256u; i += THREADS) { (&s_wh[0][0])[i] = 0u; ((unsigned long long*)s_kv)[i] = 0ull; }
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= ctl->n_tiles) return;                             // (the launch is sized by a bound; uniform over the workgroup)
    const uint32_t seg = tile_seg[tile], tfirst = tile_first[seg];
    const uint32_t seg_lo = seg_off[seg];
    const uint32_t tile_base = seg_lo + (tile - tfirst) * TILE;
    const uint32_t n_valid = min(TILE, seg_off[seg + 1] - tile_base);
    if (lb_next && own) lb_next[(size_t)tile * 256u + t] = 0u;    // the next pass's look-back words (its tiles are these tiles)
    // ---- load: wavefront w takes 64 * KPT consecutive pairs, slot j of lane l = w * 64 * KPT + j * 64 + l ----
    uint32_t key[KPT], val[KPT];
    const uint32_t wbase = w * (64u * KPT) + lane;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t s = wbase + (uint32_t)j * 64u;
        key[j] = s < n_valid ? kin[tile_base + s] : 0xFFFFFFFFu;  // beyond the end: the largest digit, ranked last, never written
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t s = wbase + (uint32_t)j * 64u;
        val[j] = s < n_valid ? vin[tile_base + s] : 0u;
    }
    // ---- rank inside the wavefront: lanes with the same digit, in lane order, behind the same digit's earlier slots.  The lanes of a slot
    //      that share a digit are found by peeling: the digit of the first lane not yet placed, one ballot, twice -- which settles a slot
    //      outright on the long runs of one cell the lists are made of; the lanes still left find each other through one 64-bit word per
    //      digit in LDS (every lane ORs its bit in and reads the word back; the set's first lane clears it).  One wavefront executes its LDS
    //      instructions in order, so no barrier is needed.  (One ballot per digit bit took ~90 instructions per key.) ----
    uint32_t rank[KPT];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = (key[j] >> shift) & dmask;
        unsigned long long rem = ~0ull, m = 0ull;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (!rem) break;
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)__builtin_ctzll(rem));
            const unsigned long long mk = __ballot(d == d0) & rem;
            if (d == d0) m = mk;
            rem &= ~mk;
        }
        const bool via_lds = (rem >> lane) & 1ull;
        if (rem) {                                                // (uniform)
            if (via_lds) atomicOr(&s_m[d], 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            if (via_lds) m = s_m[d];
        }
        const uint32_t c0 = s_wh[w][d];                          // (every lane reads before the set's first lane writes)
        __builtin_amdgcn_wave_barrier();
        if ((m & lt) == 0ull) { s_wh[w][d] = c0 + (uint32_t)__popcll(m); if (via_lds) s_m[d] = 0ull; }
        __builtin_amdgcn_wave_barrier();
        rank[j] = c0 + (uint32_t)__popcll(m & lt);
    }
    __syncthreads();
    // ---- digit t: counts per wavefront -> first slot per wavefront; tile-wide and segment-wide exclusive digit prefixes ----
    uint32_t cw[NW], tot = 0, gcount = 0, ti = 0, gi = 0, loff = 0, gbase = 0;
    if (own) {
#pragma unroll
        for (int x = 0; x < NW; ++x) { cw[x] = s_wh[x][t]; tot += cw[x]; }
        __hip_atomic_store(lb + (size_t)tile * 256u + t, LB_PARTIAL | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // at once: the tiles behind look for it
        gcount = hist[((size_t)pass * n_seg + seg) * 256u + t];
        ti = tot; gi = gcount;                                    // inclusive scans over the 256 digits (wavefront, then the four totals)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(ti, o, 64), b = __shfl_up(gi, o, 64);
            if (lane >= (uint32_t)o) { ti += a; gi += b; }
        }
        if (lane == 63u) { s_part[0][w] = ti; s_part[1][w] = gi; }
    }
    __syncthreads();
    if (own) {
        uint32_t tpre = 0, gpre = 0;
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x) if (x < w) { tpre += s_part[0][x]; gpre += s_part[1][x]; }
        loff = tpre + ti - tot; gbase = seg_lo + gpre + gi - gcount;        // exclusive; the segment's keys stay in the segment
        uint32_t run = loff;
#pragma unroll
        for (int x = 0; x < NW; ++x) { s_wh[x][t] = run; run += cw[x]; }
    }
    __syncthreads();
    // ---- reorder through LDS (the ranks need no other tile) ----
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = (key[j] >> shift) & dmask;
        const uint32_t pos = s_wh[w][d] + rank[j];
        s_k[pos] = key[j]; s_v[pos] = val[j];
    }
    // ---- decoupled look-back of digit t over the earlier tiles of this segment: four predecessors are requested at a time (independent
    //      loads), taken in order, up to the first tile that already knows its inclusive prefix ----
    if (own) {
        uint32_t excl = 0;
        const unsigned long long t_start = wall_clock64();
        bool done = false;
        for (uint32_t back = tile; back > tfirst && !done;) {
            uint32_t sv[4];
            const uint32_t nb = min(4u, back - tfirst);
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u)
                sv[u] = u < nb ? __hip_atomic_load(lb + (size_t)(back - 1u - u) * 256u + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_COMPLETE;
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                if (done || u >= nb) continue;
                uint32_t v = sv[u];
                while ((v & (LB_PARTIAL | LB_COMPLETE)) == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t_start > 1000000000ull) { ctl->err = 1u; v = LB_COMPLETE; break; }   // 10 s: report, do not hang
                    v = __hip_atomic_load(lb + (size_t)(back - 1u - u) * 256u + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                excl += v & LB_VALUE;
                if (v & LB_COMPLETE) done = true;
            }
            back -= nb;
        }
        __hip_atomic_store(lb + (size_t)tile * 256u + t, LB_COMPLETE | ((excl + tot) & LB_VALUE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_gpos[t] = gbase + excl - loff;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t s = t + (uint32_t)i * THREADS;
        if (s < n_valid) {
            const uint32_t k = s_k[s];
            const uint32_t o = s_gpos[(k >> shift) & dmask] + s;
            kout[o] = k; vout[o] = s_v[s];
        }
    }
}

// Segments of at most SMALL_CAP pairs -- most bases of a trial (the median base of a Cm trial keeps ~700 of its pairs), and every base of a
// small-frame batch (64 ycb trials: 6 400 bases of ~100 pairs) -- are sorted WHOLE by one workgroup in LDS: all passes in one launch, no
// histogram, no look-back, one read and one write of the segment.  Same slot order and the same ranking as the tiles of the onesweep passes.
#define SMALL_NW 4
#define SMALL_KPT 8
#define SMALL_CAP (64 * SMALL_NW * SMALL_KPT)
__global__ __launch_bounds__(64 * SMALL_NW) void seg_small_sort_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout,
                                                                       uint32_t* __restrict__ vout, SortPlan P, const uint32_t* __restrict__ seg_off, uint32_t n_seg, uint32_t n) {
    constexpr uint32_t NW = SMALL_NW, KPT = SMALL_KPT, THREADS = 64u * NW, TILE = THREADS * KPT;
    __shared__ uint32_t s_wh[NW][256];
    __shared__ uint32_t s_k[TILE], s_v[TILE];
    __shared__ unsigned long long s_mm[NW][256];
    __shared__ uint32_t s_part[4];
    const uint32_t seg = blockIdx.x, t = threadIdx.x, w = t >> 6, lane = t & 63u;
    uint32_t lo = seg_off ? seg_off[seg] : 0u, hi = seg_off ? seg_off[seg + 1] : n;
    lo = min(lo, n); hi = min(max(hi, lo), n);
    const uint32_t len = hi - lo;
    if (len == 0u || len > TILE) return;                          // (uniform) longer segments go through the tiles of the onesweep passes
    unsigned long long* const s_m = &s_mm[w][0];
    uint32_t key[KPT], val[KPT];
    const uint32_t wbase = w * (64u * KPT) + lane;
#pragma unroll
    for (int j = 0; j < (int)KPT; ++j) {
        const uint32_t sl = wbase + (uint32_t)j * 64u;
        key[j] = sl < len ? kin[lo + sl] : 0xFFFFFFFFu;           // beyond the end: the largest digit in every pass, ranked last, never written
        val[j] = sl < len ? vin[lo + sl] : 0u;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int p = 0; p < P.n_pass; ++p) {
        const int shift = P.shift[p];
        const uint32_t dmask = (1u << P.bits[p]) - 1u;
        for (uint32_t i = t; i < NW * 256u; i += THREADS) { (&s_wh[0][0])[i] = 0u; (&s_mm[0][0])[i] = 0ull; }
        __syncthreads();
        uint32_t rank[KPT];
#pragma unroll
        for (int j = 0; j < (int)KPT; ++j) {                      // (the ranking of seg_onesweep_kernel)
            const uint32_t d = (key[j] >> shift) & dmask;
            unsigned long long rem = ~0ull, m = 0ull;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                if (!rem) break;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)__builtin_ctzll(rem));
                const unsigned long long mk = __ballot(d == d0) & rem;
                if (d == d0) m = mk;
                rem &= ~mk;
            }
            const bool via_lds = (rem >> lane) & 1ull;
            if (rem) {
                if (via_lds) atomicOr(&s_m[d], 1ull << lane);
                __builtin_amdgcn_wave_barrier();
                if (via_lds) m = s_m[d];
            }
            const uint32_t c0 = s_wh[w][d];
            __builtin_amdgcn_wave_barrier();
            if ((m & lt) == 0ull) { s_wh[w][d] = c0 + (uint32_t)__popcll(m); if (via_lds) s_m[d] = 0ull; }
            __builtin_amdgcn_wave_barrier();
            rank[j] = c0 + (uint32_t)__popcll(m & lt);
        }
        __syncthreads();
        {   // digit t: counts per wavefront -> first slot per wavefront (exclusive scan over the 256 digits)
            uint32_t cw[NW], tot = 0;
#pragma unroll
            for (int x = 0; x < (int)NW; ++x) { cw[x] = s_wh[x][t]; tot += cw[x]; }
            uint32_t ti = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_up(ti, o, 64); if (lane >= (uint32_t)o) ti += a; }
            if (lane == 63u) s_part[w] = ti;
            __syncthreads();
            uint32_t run = ti - tot;
            for (uint32_t x = 0; x < w; ++x) run += s_part[x];
#pragma unroll
            for (int x = 0; x < (int)NW; ++x) { s_wh[x][t] = run; run += cw[x]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (int)KPT; ++j) {
            const uint32_t pos = s_wh[w][(key[j] >> shift) & dmask] + rank[j];
            s_k[pos] = key[j]; s_v[pos] = val[j];
        }
        __syncthreads();
        if (p + 1 < P.n_pass) {                                   // the next pass reads the slots in memory order again
#pragma unroll
            for (int j = 0; j < (int)KPT; ++j) { const uint32_t sl = wbase + (uint32_t)j * 64u; key[j] = s_k[sl]; val[j] = s_v[sl]; }
            __syncthreads();
        }
    }
    for (uint32_t sl = t; sl < len; sl += THREADS) { kout[lo + sl] = s_k[sl]; vout[lo + sl] = s_v[sl]; }
}

size_t sort_own_err_offset() { return offsetof(SortCtl, err); }

// Tile shape (NW << 8) | KPT by the size of the sort (tools/sort_bench.py, ms own / rocPRIM on random keys, unsegmented): 4 096-pair tiles
// for small lists, 8 192 from 1 M pairs on, 16 384 from 32 M on.  STOCS_SORT_SHAPE = "<wavefronts>x<pairs per thread>" overrides.
static int sort_shape(size_t n) {
    static int forced = -1;
    if (forced < 0) {
        forced = 0;
        if (const char* e = getenv("STOCS_SORT_SHAPE")) { int a = 0, b = 0; if (sscanf(e, "%dx%d", &a, &b) == 2 && (a == 4 || a == 8 || a == 16) && (b == 8 || b == 16)) forced = (a << 8) | b; }
    }
    if (forced) return forced;
    return n >= ((size_t)32 << 20) ? ((16 << 8) | 16) : (n >= ((size_t)1 << 20) ? ((8 << 8) | 16) : ((4 << 8) | 16));
}

// Calling convention of prims.h (tmp == NULL: *bytes receives the temporary size).  n: a HOST-side bound of the list's length (buffers and
// launches are sized by it); seg_off (device, n_seg + 1 ascending offsets) gives the segments -- each is sorted by key bits [b0, b1) on its
// own and stays where it is; seg_off == NULL: one segment [0, n).  kin / vin are not modified.
hipError_t sort_pairs_own(void* tmp, size_t& bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1,
                          const uint32_t* seg_off, uint32_t n_seg, hipStream_t st) {
    if (!seg_off) n_seg = 1;
    const SortPlan P = sort_plan(b0, b1);
    const int shape = sort_shape(n), nw = shape >> 8, kpt = shape & 255;
    const size_t tile = (size_t)64 * nw * kpt;
    // (only segments beyond SMALL_CAP pairs are cut into tiles: at most n / SMALL_CAP of them)
    const size_t max_tiles = n / tile + std::min<size_t>((size_t)n_seg, n / SMALL_CAP + 1) + 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t ctl_b = al(sizeof(SortCtl)), hist_b = al((size_t)P.n_pass * n_seg * 256 * 4), tf_b = al(((size_t)n_seg + 1) * 4), ts_b = al(max_tiles * 4),
                 lb_b = al(max_tiles * 256 * 4), buf_b = al(n * 4);
    const size_t need = ctl_b + hist_b + 2 * tf_b + ts_b + 2 * lb_b + (P.n_pass > 1 ? 2 * buf_b : 0);
    if (!tmp) { bytes = need; return hipSuccess; }
    if (bytes < need || n >= ((size_t)1 << 30) || b1 > 32 || b1 <= b0 || P.n_pass > SORT_MAX_PASS || n_seg >= (1u << 24)) return hipErrorInvalidValue;
    if (n == 0 || n_seg == 0) return hipSuccess;
    char* base = (char*)tmp;
    SortCtl* ctl = (SortCtl*)base;
    uint32_t* hist = (uint32_t*)(base + ctl_b);
    uint32_t* tile_first = (uint32_t*)(base + ctl_b + hist_b);
    uint32_t* own_off = (uint32_t*)(base + ctl_b + hist_b + tf_b);
    uint32_t* tile_seg = (uint32_t*)(base + ctl_b + hist_b + 2 * tf_b);
    uint32_t* lbs[2] = {(uint32_t*)(base + ctl_b + hist_b + 2 * tf_b + ts_b), (uint32_t*)(base + ctl_b + hist_b + 2 * tf_b + ts_b + lb_b)};
    uint32_t* tk = (uint32_t*)(base + ctl_b + hist_b + 2 * tf_b + ts_b + 2 * lb_b);
    uint32_t* tv = (uint32_t*)((char*)tk + buf_b);
    // (the sort's one zero fill -- tickets, error word, histograms -- rides in the tile-table launch: a launch less on the critical path of a trial)
    const uint32_t hist_words = (uint32_t)(hist_b / 4);
    hipLaunchKernelGGL(seg_tiles_kernel, dim3(1u + (hist_words + 4095u) / 4096u), dim3(1024), 0, st, seg_off, n_seg, (uint32_t)n, (uint32_t)tile, (uint32_t)max_tiles, tile_first, tile_seg, own_off, ctl,
                       hist, hist_words, (uint32_t)SMALL_CAP);
    hipLaunchKernelGGL(seg_small_sort_kernel, dim3(n_seg), dim3(64 * SMALL_NW), 0, st, kin, vin, kout, vout, P, seg_off, n_seg, (uint32_t)n);
#define SORT_SHAPES(X) switch (shape) { case (4 << 8) | 8: X(4, 8); break; case (8 << 8) | 8: X(8, 8); break; case (8 << 8) | 16: X(8, 16); break; \
                                        case (16 << 8) | 8: X(16, 8); break; case (16 << 8) | 16: X(16, 16); break; default: X(4, 16); break; }
#define SORT_HIST(NWV, KPTV) { if (hist_sub == 4) hipLaunchKernelGGL((seg_hist_kernel<NWV, KPTV, 4>), dim3((unsigned)max_tiles * 4u), dim3(64 * NWV), 0, st, kin, P, (const SortCtl*)ctl, \
                                    (const uint32_t*)tile_first, (const uint32_t*)tile_seg, (const uint32_t*)own_off, n_seg, hist, lbs[0]); \
                               else hipLaunchKernelGGL((seg_hist_kernel<NWV, KPTV, 1>), dim3((unsigned)max_tiles), dim3(64 * NWV), 0, st, kin, P, (const SortCtl*)ctl, (const uint32_t*)tile_first, \
                                    (const uint32_t*)tile_seg, (const uint32_t*)own_off, n_seg, hist, lbs[0]); }
    static const int hist_sub_env = getenv("STOCS_SORT_HIST_SUB") ? atoi(getenv("STOCS_SORT_HIST_SUB")) : 0;
    const int hist_sub = hist_sub_env ? hist_sub_env : (n < ((size_t)16 << 20) ? 4 : 1);
    SORT_SHAPES(SORT_HIST)
#undef SORT_HIST
    const uint32_t* sk = kin; const uint32_t* sv = vin;
    for (int p = 0; p < P.n_pass; ++p) {
        const bool to_out = ((P.n_pass - 1 - p) & 1) == 0;
        uint32_t* dk = to_out ? kout : tk; uint32_t* dv = to_out ? vout : tv;
        uint32_t* lbn = p + 1 < P.n_pass ? lbs[(p + 1) & 1] : (uint32_t*)NULL;
#define SORT_PASS(NWV, KPTV) hipLaunchKernelGGL((seg_onesweep_kernel<NWV, KPTV>), dim3((unsigned)max_tiles), dim3(64 * NWV), 0, st, sk, sv, dk, dv, p, P.shift[p], P.bits[p], ctl, \
                                                (const uint32_t*)tile_first, (const uint32_t*)tile_seg, (const uint32_t*)own_off, n_seg, (const uint32_t*)hist, lbs[p & 1], lbn)
        SORT_SHAPES(SORT_PASS)
#undef SORT_PASS
        sk = dk; sv = dv;
    }
#undef SORT_SHAPES
    return hipGetLastError();
}

}  // namespace stocs

using namespace stocs;

extern "C" {

// Diagnostics (tests/test_sort_gpu.py, tools/sort_bench.py): sorts n host pairs by key bits [0, end_bit) on the device with the library's own
// sort (which = 1; seg_off != NULL: every segment [seg_off[s], seg_off[s + 1]) on its own) or rocPRIM's (which = 0; unsegmented), `reps` times;
// returns the sorted pairs and the average device time of one sort (HIP events).
int stocs_debug_sort_pairs(int device, const uint32_t* keys, const uint32_t* vals, int64_t n, int end_bit, int which, int reps, uint32_t* keys_out, uint32_t* vals_out,
                           float* ms_per_sort, const uint32_t* seg_off, int n_seg) {
    if (n < 0 || end_bit < 1 || end_bit > 32 || reps < 1 || (n && (!keys || !vals)) || n_seg < 0 || (seg_off && !which)) return STOCS_ERR_INVALID;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    DeviceGuard dev_guard(device);
    uint32_t *dk = NULL, *dv = NULL, *ok = NULL, *ov = NULL, *dso = NULL; void* tmp = NULL;
    size_t tb = 0;
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    hipError_t e = which ? sort_pairs_own(NULL, tb, dk, ok, dv, ov, (size_t)n, 0, (unsigned)end_bit, seg_off, (uint32_t)n_seg, 0)
                         : sort_pairs(NULL, tb, (const uint32_t*)dk, ok, (const uint32_t*)dv, ov, (size_t)n, 0, (unsigned)end_bit, 0);
    if (e != hipSuccess) return STOCS_ERR_HIP;
    int rc = STOCS_OK;
    hipEvent_t e0 = NULL, e1 = NULL;
    hipStream_t st = NULL;
#define DS_TRY(x) do { if ((x) != hipSuccess) { set_error("stocs_debug_sort_pairs: %s", #x); rc = STOCS_ERR_HIP; goto done; } } while (0)
    DS_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    DS_TRY(hipMalloc((void**)&dk, nn * 4)); DS_TRY(hipMalloc((void**)&dv, nn * 4)); DS_TRY(hipMalloc((void**)&ok, nn * 4)); DS_TRY(hipMalloc((void**)&ov, nn * 4));
    DS_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 256)));
    DS_TRY(hipMemcpy(dk, keys, (size_t)n * 4, hipMemcpyHostToDevice)); DS_TRY(hipMemcpy(dv, vals, (size_t)n * 4, hipMemcpyHostToDevice));
    if (seg_off) { DS_TRY(hipMalloc((void**)&dso, ((size_t)n_seg + 1) * 4)); DS_TRY(hipMemcpy(dso, seg_off, ((size_t)n_seg + 1) * 4, hipMemcpyHostToDevice)); }
    DS_TRY(hipEventCreate(&e0)); DS_TRY(hipEventCreate(&e1));
    for (int r = 0; r < reps + 1; ++r) {       // (the first run is a warm-up)
        if (r == 1) DS_TRY(hipEventRecord(e0, st));
        if (n > 0) {
            if (which) DS_TRY(sort_pairs_own(tmp, tb, dk, ok, dv, ov, (size_t)n, 0, (unsigned)end_bit, dso, (uint32_t)n_seg, st));
            else DS_TRY(sort_pairs(tmp, tb, (const uint32_t*)dk, ok, (const uint32_t*)dv, ov, (size_t)n, 0, (unsigned)end_bit, st));
        }
    }
    DS_TRY(hipEventRecord(e1, st));
    DS_TRY(hipStreamSynchronize(st));
    { float ms = 0; DS_TRY(hipEventElapsedTime(&ms, e0, e1)); if (ms_per_sort) *ms_per_sort = ms / (float)reps; }
    if (which && n > 0) { uint32_t err = 0; DS_TRY(hipMemcpy(&err, (char*)tmp + sort_own_err_offset(), 4, hipMemcpyDeviceToHost)); if (err) { set_error("sort: a look-back wait ran into its bound"); rc = STOCS_ERR_HIP; goto done; } }
    if (keys_out) DS_TRY(hipMemcpy(keys_out, ok, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (vals_out) DS_TRY(hipMemcpy(vals_out, ov, (size_t)n * 4, hipMemcpyDeviceToHost));
done:
#undef DS_TRY
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(ok); (void)hipFree(ov); (void)hipFree(tmp); (void)hipFree(dso);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

}  // extern "C"
