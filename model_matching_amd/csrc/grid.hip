// grid.hip -- the scene brick grid (SceneGrid, stocs_ctx.h) built on the GPU.
// Replaces the kd-tree construction of the reference (Super4PCS::KdTree, include/super4pcs/accelerators/
// kdtree.h:191-330, built by stocs_estimator::kdtree_initialize, stocs.cpp:966-980) as the spatial index of the
// verification pass; the restricted-radius query it has to answer is kdtree.h:387-442.
//
// A new scene arrives with every camera frame, so the build is part of the per-frame latency: every step is
// a kernel or a rocPRIM primitive on the context's stream (the host only sizes buffers):
//   1. count / fill the (cell, point) incidences: a point belongs to every cell whose box is within
//      r = 1.001 eps of it (double arithmetic, the same predicate for count and fill);
//   2. one stable radix sort by cell key (brick << 9 | local cell) -- incidences are generated in ascending
//      point order, so a cell's list comes out in ascending scene index (the tie rule of the scan kernels
//      relies on it); dense scenes append a 16-bit quantised distance to the cell centre to the key;
//   3. run boundaries -> non-empty cells -> bricks (prefix sums), padded list offsets (multiples of 8
//      entries = whole 128-byte lines, sentinel-filled);
//   4. cell words (offset, count, 64-bit sub-cell mask), the top-level brick table, the lists;
//   5. dense scenes: per 8-entry chunk a lower bound of the distance to the cell centre (suffix minimum of
//      the exact distances, so the quantised order only affects how early the scan stops, never the result).
// HBM-bound integer/byte work; no MFMA.
#include <math.h>
#include <string.h>
#include <cstring>


#include <algorithm>
#include <vector>

#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

struct GridGeom {
    double of[3];     // origin (the float origin the scan kernels use, widened)
    double h, r;      // cell edge, inclusion radius
    int n[3];         // cells per axis
    int nbx, nby;
    int dense;        // 1: key carries the quantised centre distance
    double qscale;    // quantisation of the centre distance to 16 bits
};

__device__ __forceinline__ void cell_range(const GridGeom& G, const double p[3], int lo[3], int hi[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = max(0, (int)floor((p[k] - G.r - G.of[k]) / G.h));
        hi[k] = min(G.n[k] - 1, (int)floor((p[k] + G.r - G.of[k]) / G.h));
    }
}

__device__ __forceinline__ double box_dist2(const GridGeom& G, const double p[3], int cx, int cy, int cz) {
    const int cc[3] = {cx, cy, cz};
    double d2 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double b0 = G.of[k] + cc[k] * G.h, b1 = b0 + G.h;
        const double d = p[k] < b0 ? b0 - p[k] : (p[k] > b1 ? p[k] - b1 : 0.0);
        d2 += d * d;
    }
    return d2;
}

// FILL = false: number of cells within r of every point; FILL = true: their (key, point) records
// keys32 != NULL (FILL; keys of at most 32 bits): the records' keys go there as 32-bit words (the library's own sort takes those)
template <bool FILL>
__global__ __launch_bounds__(256) void incidence_kernel(GridGeom G, const float4* __restrict__ spos, int nS, uint32_t* __restrict__ cnt,
                                                        const unsigned long long* __restrict__ off, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, uint32_t* __restrict__ keys32 = NULL) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nS) return;
    const float4 pf = spos[i];
    const double p[3] = {pf.x, pf.y, pf.z};
    int lo[3], hi[3];
    cell_range(G, p, lo, hi);
    uint32_t c = 0;
    unsigned long long o = FILL ? off[i] : 0ull;
    for (int cz = lo[2]; cz <= hi[2]; ++cz)
        for (int cy = lo[1]; cy <= hi[1]; ++cy)
            for (int cx = lo[0]; cx <= hi[0]; ++cx) {
                if (box_dist2(G, p, cx, cy, cz) > G.r * G.r) continue;
                if (FILL) {
                    const uint64_t brick = ((uint64_t)(cz >> 3) * G.nby + (uint64_t)(cy >> 3)) * G.nbx + (uint64_t)(cx >> 3);
                    uint64_t key = (brick << 9) | (uint64_t)(((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7));
                    if (G.dense) {
                        const double dx = p[0] - (G.of[0] + (cx + 0.5) * G.h), dy = p[1] - (G.of[1] + (cy + 0.5) * G.h),
                                     dz = p[2] - (G.of[2] + (cz + 0.5) * G.h);
                        const double q = sqrt(dx * dx + dy * dy + dz * dz) * G.qscale;
                        key = (key << 16) | (uint64_t)(q < 65535.0 ? (unsigned)q : 65535u);
                    }
                    if (keys32) keys32[o + c] = (uint32_t)key; else keys[o + c] = key;
                    vals[o + c] = (uint32_t)i;
                }
                c++;
            }
    if (!FILL) cnt[i] = c;
}

__global__ __launch_bounds__(256) void widen_keys_kernel(const uint32_t* __restrict__ k32, size_t n, uint64_t* __restrict__ k64) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) k64[e] = (uint64_t)k32[e];
}

// first incidence of every cell (keys sorted; the low `qbits` bits are not part of the cell)
__global__ __launch_bounds__(256) void cell_flags_kernel(const uint64_t* __restrict__ keys, size_t n, int qbits, uint32_t* __restrict__ cflag,
                                                         uint32_t* __restrict__ bflag) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint64_t ck = keys[e] >> qbits;
    const uint64_t pk = e ? (keys[e - 1] >> qbits) : ~0ull;
    cflag[e] = (e == 0 || ck != pk) ? 1u : 0u;
    bflag[e] = (e == 0 || (ck >> 9) != (pk >> 9)) ? 1u : 0u;   // first incidence of a brick
}

// per non-empty cell: first incidence, cell key, brick ordinal
__global__ __launch_bounds__(256) void cell_records_kernel(const uint64_t* __restrict__ keys, size_t n, int qbits, const uint32_t* __restrict__ cflag,
                                                           const uint32_t* __restrict__ cidx, const uint32_t* __restrict__ bflag,
                                                           const uint32_t* __restrict__ bidx, uint32_t* __restrict__ cell_first, uint64_t* __restrict__ cell_key,
                                                           uint32_t* __restrict__ cell_brick) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || !cflag[e]) return;
    const uint32_t c = cidx[e];        // exclusive scan of cflag = ordinal of this cell
    cell_first[c] = (uint32_t)e;
    cell_key[c] = keys[e] >> qbits;
    // bidx = exclusive scan of the brick-start flags: the bricks that start strictly before e.  The cell's own brick
    // is that many if e opens it, one less otherwise
    cell_brick[c] = bidx[e] + bflag[e] - 1u;
}

// padded length of every cell's list
__global__ __launch_bounds__(256) void cell_padded_kernel(const uint32_t* __restrict__ cell_first, uint32_t n_cells, uint32_t n_inc, uint32_t pad,
                                                          uint32_t* __restrict__ padded, uint32_t* __restrict__ max_count) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const uint32_t cnt = (c + 1 < n_cells ? cell_first[c + 1] : n_inc) - cell_first[c];
    padded[c] = (cnt + pad - 1u) & ~(pad - 1u);   // whole 128-byte lines: 8 entries of 16 bytes, or 16 of 8 bytes (dense scenes)
    atomicMax(max_count, cnt);
}

__global__ __launch_bounds__(256) void fill_i32_kernel(int32_t* __restrict__ a, size_t n, int32_t v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}
__global__ __launch_bounds__(256) void fill_list_kernel(float4* __restrict__ a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = make_float4(1.0e30f, 1.0e30f, 1.0e30f, __int_as_float(-1));   // sentinel: far away, index -1
}
__global__ __launch_bounds__(256) void zero_cells_kernel(uint4* __restrict__ a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = make_uint4(0u, 0u, 0u, 0u);
}

// cell word + top table; one wavefront per non-empty cell, lane = sub-cell.  Sub-cell masks (cell edge = eps
// only): bit s is set when some point of the cell's list lies within r of sub-cell s (any point that close to a
// sub-cell is that close to the cell, hence in its list); the 64 lanes ballot the mask.
__global__ __launch_bounds__(256) void cell_words_kernel(GridGeom G, int div, const uint32_t* __restrict__ cell_first, const uint64_t* __restrict__ cell_key,
                                                         const uint32_t* __restrict__ cell_brick, const uint32_t* __restrict__ list_off,
                                                         uint32_t n_cells, uint32_t n_inc, const uint32_t* __restrict__ vals,
                                                         const float4* __restrict__ spos, int32_t* __restrict__ top, uint4* __restrict__ cells,
                                                         uint4* __restrict__ flat, const uint32_t* __restrict__ kept) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int s = threadIdx.x & 63;
    if (c >= n_cells) return;   // whole wavefront
    const uint32_t first = cell_first[c];
    const uint32_t cnt = (c + 1 < n_cells ? cell_first[c + 1] : n_inc) - first;
    const uint64_t key = cell_key[c];
    const uint64_t brick = key >> 9;
    const uint32_t local = (uint32_t)(key & 511);
    const uint32_t b = cell_brick[c];
    uint32_t mlo = 0xFFFFFFFFu, mhi = 0xFFFFFFFFu;
    if (div == 1) {
        const int bx = (int)(brick % (uint64_t)G.nbx), by = (int)((brick / (uint64_t)G.nbx) % (uint64_t)G.nby),
                  bz = (int)(brick / ((uint64_t)G.nbx * (uint64_t)G.nby));
        const int cx = bx * 8 + (int)(local & 7), cy = by * 8 + (int)((local >> 3) & 7), cz = bz * 8 + (int)(local >> 6);
        const double hs = G.h / 4.0;
        const int sc[3] = {4 * cx + (s & 3), 4 * cy + ((s >> 2) & 3), 4 * cz + (s >> 4)};
        bool any = false;
        for (uint32_t k = 0; k < cnt && !any; ++k) {
            const float4 pf = spos[vals[first + k]];   // same address in every lane: one broadcast load
            const double p[3] = {pf.x, pf.y, pf.z};
            double d2 = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double b0 = G.of[a] + sc[a] * hs, b1 = b0 + hs;
                const double d = p[a] < b0 ? b0 - p[a] : (p[a] > b1 ? p[a] - b1 : 0.0);
                d2 += d * d;
            }
            any = d2 <= G.r * G.r;
        }
        const unsigned long long m = __ballot(any);
        mlo = (uint32_t)(m & 0xFFFFFFFFull); mhi = (uint32_t)(m >> 32);
    }
    if (s == 0) {
        top[brick] = (int32_t)b;   // every cell of the brick writes the same value
        const uint4 w = make_uint4(list_off[c], kept ? kept[c] : cnt, mlo, mhi);   // (pruned lists: the mask above still comes from every point within r)
        cells[(size_t)b * 512 + local] = w;
        if (flat) {
            const int bx = (int)(brick % (uint64_t)G.nbx), by = (int)((brick / (uint64_t)G.nbx) % (uint64_t)G.nby), bz = (int)(brick / ((uint64_t)G.nbx * (uint64_t)G.nby));
            const int cx = bx * 8 + (int)(local & 7), cy = by * 8 + (int)((local >> 3) & 7), cz = bz * 8 + (int)(local >> 6);
            if (cx < G.n[0] && cy < G.n[1] && cz < G.n[2]) flat[((size_t)cz * G.n[1] + cy) * G.n[0] + cx] = w;
        }
    }
}

// list entries: incidence e of cell c goes to list_off[c] + (e - first[c])
__global__ __launch_bounds__(256) void list_fill_kernel(const uint32_t* __restrict__ cflag, const uint32_t* __restrict__ cidx,
                                                        const uint32_t* __restrict__ cell_first, const uint32_t* __restrict__ list_off,
                                                        const uint32_t* __restrict__ vals, size_t n, const float4* __restrict__ spos,
                                                        float4* __restrict__ list, const uint32_t* __restrict__ rank) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (rank) {                        // pruned lists: the survivors of a cell, in order
        const uint32_t r = rank[e];
        if (r == 0xFFFFFFFFu) return;
        const uint32_t c = cidx[e] + cflag[e] - 1u;
        const uint32_t pt = vals[e];
        const float4 p = spos[pt];
        list[(size_t)list_off[c] + r] = make_float4(p.x, p.y, p.z, __int_as_float((int)pt));
        return;
    }
    // cidx = exclusive scan of the cell-start flags: the cells that start strictly before e; the own cell is that
    // many if e opens it, one less otherwise
    const uint32_t c = cidx[e] + cflag[e] - 1u;
    const uint32_t pt = vals[e];
    const float4 p = spos[pt];
    list[(size_t)list_off[c] + (e - cell_first[c])] = make_float4(p.x, p.y, p.z, __int_as_float((int)pt));
}

// dense scenes: chunk_r[j] = (minimum exact distance to the cell centre over chunks >= j), rounded down
__global__ __launch_bounds__(256) void chunk_bounds_kernel(GridGeom G, const uint32_t* __restrict__ cell_first, const uint64_t* __restrict__ cell_key,
                                                           const uint32_t* __restrict__ list_off, uint32_t n_cells, uint32_t n_inc,
                                                           const float4* __restrict__ list, float* __restrict__ chunk_r) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const uint32_t cnt = (c + 1 < n_cells ? cell_first[c + 1] : n_inc) - cell_first[c];
    const uint64_t key = cell_key[c];
    const uint64_t brick = key >> 9;
    const uint32_t local = (uint32_t)(key & 511);
    const int bx = (int)(brick % (uint64_t)G.nbx), by = (int)((brick / (uint64_t)G.nbx) % (uint64_t)G.nby),
              bz = (int)(brick / ((uint64_t)G.nbx * (uint64_t)G.nby));
    const int cx = bx * 8 + (int)(local & 7), cy = by * 8 + (int)((local >> 3) & 7), cz = bz * 8 + (int)(local >> 6);
    const double ccx = G.of[0] + (cx + 0.5) * G.h, ccy = G.of[1] + (cy + 0.5) * G.h, ccz = G.of[2] + (cz + 0.5) * G.h;
    const uint32_t o = list_off[c];
    double run = 1e300;
    for (int k = (int)cnt - 1; k >= 0; --k) {
        const float4 e = list[(size_t)o + k];
        const double dx = e.x - ccx, dy = e.y - ccy, dz = e.z - ccz;
        run = fmin(run, sqrt(dx * dx + dy * dy + dz * dz));
        if ((k & 7) == 0) chunk_r[(o + (uint32_t)k) >> 3] = (float)(run - (2e-6 + 2.5e-7 * run));   // below the exact value also after the conversion to float, at any coordinate scale
    }
}

// dense scenes with cell edges below epsilon (the sub-cell masks are all ones there): the z word of a cell takes a lower bound of
// the distance from the cell centre to its nearest listed point (= the bound of the list's first chunk).  A query at distance t
// from the centre has no neighbour within epsilon when bound - t > epsilon, and the scan kernel then never touches the list.
__global__ __launch_bounds__(256) void cell_nearest_kernel(const uint64_t* __restrict__ cell_key, const uint32_t* __restrict__ cell_brick, const uint32_t* __restrict__ list_off,
                                                           uint32_t n_cells, const float* __restrict__ chunk_r, uint4* __restrict__ cells) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    cells[(size_t)cell_brick[c] * 512 + (uint32_t)(cell_key[c] & 511)].z = __float_as_uint(chunk_r[list_off[c] >> 3]);
}


// ---------------------------------------------------------------------------------------------
// Dominance pruning of the candidate lists (round 5).  A cell's list holds every scene point within r of the cell box, so that ONE
// list answers the radius-epsilon nearest-neighbour query of any position in the cell; but most of those points can never be the
// ANSWER for a position in this cell.  For two listed points p, p' the difference of the squared distances from a position q is
// linear in q,
//     f(q) = |q - p|^2 - |q - p'|^2 = |u|^2 - |u'|^2 - 2 s . (u - u'),      u = p - centre, u' = p' - centre, s = q - centre,
// so over the (slightly widened) cell box |s_k| <= hh its minimum is |u|^2 - |u'|^2 - 2 hh |u - u'|_1.  When that minimum exceeds a
// margin that covers the scan kernels' float evaluation of both squared distances, p' is strictly nearer than p for EVERY position
// the kernels can assign to this cell: p never wins there -- not on distance, not on a tie -- and is dropped from the list.  The
// nearest neighbour of every query, its index and the tie rule are unchanged (whoever wins a query is by definition not dominated);
// the kd-tree of the reference (kdtree.h:394-459) returns that same point.  Dominators tried per entry: the PRUNE_K listed points
// nearest to the cell centre (any listed point may serve; these prune best).  On a surface sampled every 1.5 mm with cells of
// 1.25 mm the lists shrink from ~45 entries to ~5 whatever the cell's height above the surface: one 128-byte line answers a query,
// where the centre-sorted lists with early exit of rounds 2-4 read 15-20 entries in a dependent line -> bound -> line chain.
// One wavefront per cell; double arithmetic relative to the cell centre (float positions are exact in double).
// ---------------------------------------------------------------------------------------------
// GS lanes per cell (64, or 16 on sparse scenes whose lists hold a handful of points: four cells per wavefront); a list longer than GS is
// walked GS entries at a time, its first GS entries stay in registers for the selection rounds.
template <int PRUNE_K, int GS>
__global__ __launch_bounds__(256) void prune_kernel(GridGeom G, double hh, double margin, const uint32_t* __restrict__ cell_first, const uint64_t* __restrict__ cell_key,
                                                    uint32_t n_cells, uint32_t n_inc, const uint32_t* __restrict__ vals, const float4* __restrict__ spos,
                                                    uint32_t* __restrict__ rank, uint32_t* __restrict__ kept, float* __restrict__ nearest) {
    const uint32_t c = blockIdx.x * (256u / GS) + (threadIdx.x / GS);
    const int lane = threadIdx.x & (GS - 1), wlane = threadIdx.x & 63;
    const bool cell_ok = c < n_cells;                             // (a wavefront of 16-lane groups may hold cells beyond the end: they idle through the ballots)
    const uint32_t first = cell_ok ? cell_first[c] : 0u;
    const uint32_t cnt = cell_ok ? (c + 1 < n_cells ? cell_first[c + 1] : n_inc) - first : 0u;
    const uint64_t key = cell_ok ? cell_key[c] : 0ull;
    const uint64_t brick = key >> 9;
    const uint32_t local = (uint32_t)(key & 511);
    const int bx = (int)(brick % (uint64_t)G.nbx), by = (int)((brick / (uint64_t)G.nbx) % (uint64_t)G.nby), bz = (int)(brick / ((uint64_t)G.nbx * (uint64_t)G.nby));
    const int cx = bx * 8 + (int)(local & 7), cy = by * 8 + (int)((local >> 3) & 7), cz = bz * 8 + (int)(local >> 6);
    const double ccx = G.of[0] + (cx + 0.5) * G.h, ccy = G.of[1] + (cy + 0.5) * G.h, ccz = G.of[2] + (cz + 0.5) * G.h;
    // the group's first GS entries, one per lane
    double ux0 = 0, uy0 = 0, uz0 = 0, n20 = 0;
    if ((uint32_t)lane < cnt) {
        const float4 pf = spos[vals[first + (uint32_t)lane]];
        ux0 = (double)pf.x - ccx; uy0 = (double)pf.y - ccy; uz0 = (double)pf.z - ccz;
        n20 = ux0 * ux0 + uy0 * uy0 + uz0 * uz0;
    }
    // ---- the PRUNE_K entries nearest to the centre: PRUNE_K rounds of "smallest (distance, position) key above the last one" ----
    double dux[PRUNE_K], duy[PRUNE_K], duz[PRUNE_K], dn2[PRUNE_K];
    unsigned long long last = 0ull;
    int nd = 0;
    uint32_t cnt_max = cnt;                                       // (uniform over the wavefront: the loops below run for the longest list of its groups)
    if (GS < 64) for (int off = GS; off < 64; off <<= 1) cnt_max = max(cnt_max, (uint32_t)__shfl_xor((int)cnt_max, off, 64));
#pragma unroll
    for (int j = 0; j < PRUNE_K; ++j) {
        if ((uint32_t)j >= cnt_max) break;
        unsigned long long best = ~0ull;
        if ((uint32_t)lane < cnt) {
            const unsigned long long kk = ((unsigned long long)__float_as_uint((float)n20) << 32) | (unsigned long long)(lane + 1);
            if (kk > last) best = kk;
        }
        for (uint32_t k = (uint32_t)lane + GS; k < cnt; k += GS) {
            const float4 pf = spos[vals[first + k]];
            const double ux = (double)pf.x - ccx, uy = (double)pf.y - ccy, uz = (double)pf.z - ccz;
            const float d2 = (float)(ux * ux + uy * uy + uz * uz);        // (selection only: any listed point is a valid dominator)
            const unsigned long long kk = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(k + 1u);
            if (kk > last && kk < best) best = kk;
        }
        for (int off = GS / 2; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(best, off, GS); best = o < best ? o : best; }
        if (best == ~0ull) continue;                              // this group's list is exhausted (another group's may not be)
        last = best;
        const uint32_t kb = (uint32_t)(best & 0xFFFFFFFFull) - 1u;
        double vx, vy, vz;
        if (kb < (uint32_t)GS) {                                  // held by a lane of the group
            vx = __shfl(ux0, (int)kb, GS); vy = __shfl(uy0, (int)kb, GS); vz = __shfl(uz0, (int)kb, GS);
        } else {
            const float4 pf = spos[vals[first + kb]];             // (the same address in every lane of the group)
            vx = (double)pf.x - ccx; vy = (double)pf.y - ccy; vz = (double)pf.z - ccz;
        }
        dux[j] = vx; duy[j] = vy; duz[j] = vz;
        dn2[j] = vx * vx + vy * vy + vz * vz;
        nd = j + 1;
    }
    // ---- every entry against the dominators ----
    uint32_t base = 0;
    for (uint32_t k0 = 0; k0 < cnt_max; k0 += GS) {
        const uint32_t k = k0 + (uint32_t)lane;
        bool keep = false;
        if (k < cnt) {
            double ux = ux0, uy = uy0, uz = uz0, n2 = n20;
            if (k0) {
                const float4 pf = spos[vals[first + k]];
                ux = (double)pf.x - ccx; uy = (double)pf.y - ccy; uz = (double)pf.z - ccz;
                n2 = ux * ux + uy * uy + uz * uz;
            }
            keep = true;
#pragma unroll
            for (int j = 0; j < PRUNE_K; ++j) {
                if (j < nd) {
                    const double fmin_ = (n2 - dn2[j]) - 2.0 * hh * (fabs(ux - dux[j]) + fabs(uy - duy[j]) + fabs(uz - duz[j]));
                    if (fmin_ > margin) keep = false;      // (an entry is never its own dominator: f is 0 there)
                }
            }
        }
        unsigned long long m = __ballot(keep);
        if (GS < 64) m = (m >> (wlane & ~(GS - 1))) & ((1ull << GS) - 1ull);      // the group's own lanes
        if (k < cnt) rank[first + k] = keep ? base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
        base += (uint32_t)__popcll(m);
    }
    if (lane == 0 && cell_ok) {
        kept[c] = base;
        // lower bound of |centre - nearest listed point| (the nearest is never dominated), below the exact value also as a float
        const double d0 = nd ? sqrt(dn2[0]) : 0.0;
        nearest[c] = (float)fmax(d0 - (2e-6 + 2.5e-7 * d0), 0.0);
    }
}

__global__ __launch_bounds__(256) void cell_padded_kept_kernel(const uint32_t* __restrict__ kept, uint32_t n_cells, uint32_t pad, uint32_t* __restrict__ padded,
                                                               uint32_t* __restrict__ max_count, unsigned long long* __restrict__ total) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const uint32_t cnt = kept[c];
    padded[c] = (cnt + pad - 1u) & ~(pad - 1u);
    atomicMax(max_count, cnt);
    unsigned long long t = cnt;
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(total, t);
}

__global__ __launch_bounds__(256) void cell_nearest_store_kernel(const uint64_t* __restrict__ cell_key, const uint32_t* __restrict__ cell_brick, uint32_t n_cells,
                                                                 const float* __restrict__ nearest, uint4* __restrict__ cells) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    cells[(size_t)cell_brick[c] * 512 + (uint32_t)(cell_key[c] & 511)].z = __float_as_uint(nearest[c]);
}

struct Tmp {   // temporaries of one build: the context's workspace arena (no hipMalloc / hipFree in steady state)
    Arena* ws;
    template <class T>
    int get(T** p, size_t n) { return ws->take(std::max<size_t>(n, 1) * sizeof(T), (void**)p); }
};

static inline unsigned grid_of(size_t n) { return (unsigned)((n + 255) / 256); }

// Builds c->grid for cell edge eps / div from c->d_spos (device) and c->h_spos (bounding box only).
// dense != 0: lists ordered by distance to the cell centre + chunk bounds (scan kernels with early exit).
int build_grid_gpu(stocs_ctx* c, int div, int dense, int prune) {
    if (prune) dense = 0;      // pruned lists stay in index order (the `<=` tie rule of the index-ordered scan) and need no early exit
    SceneGrid& g = c->grid;
    const int nS = c->nS;
    const double eps = (double)c->prm.distance_threshold;
    if (div < 1 || div > 4) div = 1;
    const double h = eps / div;
    const double r = eps * 1.001;  // safety margin >> float rounding of the device cell computation
    double mn[3] = {1e30, 1e30, 1e30}, mx[3] = {-1e30, -1e30, -1e30};
    for (int i = 0; i < nS; ++i) {
        const V3 p = c->h_spos[i];
        const double v[3] = {p.x, p.y, p.z};
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[k]); mx[k] = std::max(mx[k], v[k]); }
    }
    if (nS == 0) { for (int k = 0; k < 3; ++k) { mn[k] = 0; mx[k] = 0; } }
    for (int k = 0; k < 3; ++k) { g.bb_mn[k] = mn[k]; g.bb_mx[k] = mx[k]; }
    g.d_dist = NULL; g.dist_ready = false;
    const double pad = r + 2 * h;  // a query outside the grid is farther than epsilon from every point
    const double o[3] = {mn[0] - pad, mn[1] - pad, mn[2] - pad};
    int n[3];
    for (int k = 0; k < 3; ++k) n[k] = (int)floor((mx[k] + pad - o[k]) / h) + 1;
    g.ox = (float)o[0]; g.oy = (float)o[1]; g.oz = (float)o[2];
    g.inv_h = (float)(1.0 / h);
    g.nx = n[0]; g.ny = n[1]; g.nz = n[2];
    g.nbx = (n[0] + 7) / 8; g.nby = (n[1] + 7) / 8; g.nbz = (n[2] + 7) / 8;
    g.h = (float)h;
    g.n_bricks = 0; g.n_entries = 0; g.avg_list_len = 0; g.has_nearest = false;
    g.d_top = NULL; g.d_cells = NULL; g.d_list = NULL; g.d_chunk_r = NULL; g.d_flat = NULL;
    const int64_t n_top = (int64_t)g.nbx * g.nby * g.nbz;
    if (n_top > (int64_t)400 * 1000 * 1000) { set_error("scene extent too large for the brick grid"); return STOCS_ERR_INVALID; }
    int cell_bits = 1;
    while (((int64_t)1 << cell_bits) < n_top * 512) cell_bits++;

    GridGeom G;
    // the scan kernels compute the cell as floor((q - o_f) * inv_h) with the FLOAT origin: use exactly that origin
    G.of[0] = g.ox; G.of[1] = g.oy; G.of[2] = g.oz;
    G.h = h; G.r = r;
    G.n[0] = n[0]; G.n[1] = n[1]; G.n[2] = n[2];
    G.nbx = g.nbx; G.nby = g.nby;
    G.dense = dense ? 1 : 0;
    G.qscale = 65535.0 / (r + h * 0.8660254037844387 + 1e-9);   // a listed point is at most r + half a cell diagonal from the centre
    const int qbits = dense ? 16 : 0;

    hipStream_t st = c->stream;
    Tmp T;
    T.ws = &c->grid_ws;
    int rc;
    STOCS_HIP_CHECK(hipStreamSynchronize(st));   // the previous grid and the previous build's temporaries are recycled
    if ((rc = c->grid_ws.reset()) || (rc = c->grid_mem.reset())) return rc;
    if ((rc = c->grid_mem.take((size_t)std::max<int64_t>(n_top, 1) * 4, (void**)&g.d_top))) return rc;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_of((size_t)n_top)), dim3(256), 0, st, g.d_top, (size_t)n_top, -1);
    if (nS == 0) {
        if ((rc = c->grid_mem.take(16, (void**)&g.d_cells)) || (rc = c->grid_mem.take(8 * 16, (void**)&g.d_list))) return rc;
        STOCS_HIP_CHECK(hipStreamSynchronize(st));
        return STOCS_OK;
    }
    // ---- 1. incidences ----
    uint32_t* d_cnt; unsigned long long* d_off;
    if ((rc = T.get(&d_cnt, (size_t)nS + 1)) || (rc = T.get(&d_off, (size_t)nS + 1))) return rc;
    hipLaunchKernelGGL(incidence_kernel<false>, dim3(grid_of(nS)), dim3(256), 0, st, G, c->d_spos, nS, d_cnt, (const unsigned long long*)NULL,
                       (uint64_t*)NULL, (uint32_t*)NULL);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)(d_cnt + nS), (size_t)1, 0);
    size_t tb = 0;
    STOCS_HIP_CHECK(exclusive_scan(NULL, tb, d_cnt, d_off, (size_t)nS + 1, st));
    char* d_tmp;
    if ((rc = T.get(&d_tmp, tb))) return rc;
    STOCS_HIP_CHECK(exclusive_scan(d_tmp, tb, d_cnt, d_off, (size_t)nS + 1, st));
    // the build's small read-backs land in a pinned slot of the context (a copy into a pageable stack word takes the runtime's staging path)
    if ((rc = ensure_pinned(c, PIN_VAR))) return rc;
    unsigned long long* rb64 = (unsigned long long*)((char*)c->h_pin + PIN_BEST + 128);   // [0] incidences, [1] kept entries
    uint32_t* rb32 = (uint32_t*)((char*)c->h_pin + PIN_BEST + 160);                     // [0..1] cells, bricks, [2] sort error, [3..4] list entries, longest list
    rb64[0] = 0; rb64[1] = 0; for (int k = 0; k < 5; ++k) rb32[k] = 0;
    STOCS_HIP_CHECK(hipMemcpyAsync(&rb64[0], d_off + nS, 8, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    const unsigned long long n_inc64 = rb64[0];
    // padded lists stay below 2^31 entries (32-bit offsets in the cell words)
    if (n_inc64 >= (1ull << 28)) { set_error("scene grid lists too large (%llu incidences)", n_inc64); return STOCS_ERR_INVALID; }
    const size_t n_inc = (size_t)n_inc64;
    uint64_t *d_keys, *d_keys_s; uint32_t *d_vals, *d_vals_s;
    if ((rc = T.get(&d_keys, n_inc)) || (rc = T.get(&d_keys_s, n_inc)) || (rc = T.get(&d_vals, n_inc)) || (rc = T.get(&d_vals_s, n_inc))) return rc;
    // ---- 2. stable sort by cell (and quantised centre distance) ----
    size_t ts = 0;
    const unsigned end_bit = (unsigned)std::min(64, cell_bits + qbits);
    // keys of at most 32 bits (every sparse scene: ~21 bits of cell) and a frame's worth of incidences: the library's own onesweep (sort32.hip,
    // one launch per 8-bit pass) where rocPRIM's 64-bit radix_sort_pairs runs ~17 small launches (150 us of a frame's stocs_ctx_set_scene in
    // round 5a); both stable, the sorted keys are widened back for the kernels behind
    const bool own_sort = end_bit <= 32 && n_inc >= 65536 && n_inc < ((size_t)1 << 30) && !(getenv("STOCS_SORT") && !strcmp(getenv("STOCS_SORT"), "rocprim"));
    char* d_ts;
    const uint32_t* d_sort_err = NULL;
    if (own_sort) {
        uint32_t* k32 = (uint32_t*)d_keys;                         // (the unsorted 64-bit array is not needed: its first half holds the 32-bit keys)
        uint32_t* k32s = k32 + n_inc;
        hipLaunchKernelGGL(incidence_kernel<true>, dim3(grid_of(nS)), dim3(256), 0, st, G, c->d_spos, nS, (uint32_t*)NULL, d_off, (uint64_t*)NULL, d_vals, k32);
        STOCS_HIP_CHECK(hipGetLastError());
        STOCS_HIP_CHECK(sort_pairs_own(NULL, ts, k32, k32s, d_vals, d_vals_s, n_inc, 0, end_bit, NULL, 1, st));
        if ((rc = T.get(&d_ts, ts))) return rc;
        STOCS_HIP_CHECK(sort_pairs_own(d_ts, ts, k32, k32s, d_vals, d_vals_s, n_inc, 0, end_bit, NULL, 1, st));
        hipLaunchKernelGGL(widen_keys_kernel, dim3(grid_of(n_inc)), dim3(256), 0, st, (const uint32_t*)k32s, n_inc, d_keys_s);
        d_sort_err = (const uint32_t*)(d_ts + sort_own_err_offset());
    } else {
    hipLaunchKernelGGL(incidence_kernel<true>, dim3(grid_of(nS)), dim3(256), 0, st, G, c->d_spos, nS, (uint32_t*)NULL, d_off, d_keys, d_vals);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(sort_pairs(NULL, ts, d_keys, d_keys_s, d_vals, d_vals_s, n_inc, 0, end_bit, st));
    if ((rc = T.get(&d_ts, ts))) return rc;
    STOCS_HIP_CHECK(sort_pairs(d_ts, ts, d_keys, d_keys_s, d_vals, d_vals_s, n_inc, 0, end_bit, st));
    }
    // ---- 3. cells and bricks ----
    uint32_t *d_cflag, *d_bflag, *d_cidx, *d_bidx;
    if ((rc = T.get(&d_cflag, n_inc + 1)) || (rc = T.get(&d_bflag, n_inc + 1)) || (rc = T.get(&d_cidx, n_inc + 1)) || (rc = T.get(&d_bidx, n_inc + 1))) return rc;
    hipLaunchKernelGGL(cell_flags_kernel, dim3(grid_of(n_inc)), dim3(256), 0, st, d_keys_s, n_inc, qbits, d_cflag, d_bflag);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)(d_cflag + n_inc), (size_t)1, 0);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)(d_bflag + n_inc), (size_t)1, 0);
    size_t t32 = 0;
    STOCS_HIP_CHECK(exclusive_scan(NULL, t32, d_cflag, d_cidx, (size_t)n_inc + 1, st));
    char* d_t32;
    if ((rc = T.get(&d_t32, t32))) return rc;
    STOCS_HIP_CHECK(exclusive_scan(d_t32, t32, d_cflag, d_cidx, (size_t)n_inc + 1, st));
    STOCS_HIP_CHECK(exclusive_scan(d_t32, t32, d_bflag, d_bidx, (size_t)n_inc + 1, st));
    uint32_t* counts = rb32;
    STOCS_HIP_CHECK(hipMemcpyAsync(&counts[0], d_cidx + n_inc, 4, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(&counts[1], d_bidx + n_inc, 4, hipMemcpyDeviceToHost, st));
    if (d_sort_err) STOCS_HIP_CHECK(hipMemcpyAsync(&rb32[2], d_sort_err, 4, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t sort_err = rb32[2];
    if (sort_err) { set_error("scene grid: the incidence sort gave up waiting for a tile (sort32.hip)"); return STOCS_ERR_HIP; }
    const uint32_t n_cells = counts[0], n_bricks = counts[1];
    uint32_t *d_cell_first, *d_cell_brick, *d_padded, *d_list_off, *d_max;
    uint64_t* d_cell_key;
    if ((rc = T.get(&d_cell_first, (size_t)n_cells + 1)) || (rc = T.get(&d_cell_brick, (size_t)n_cells + 1)) || (rc = T.get(&d_padded, (size_t)n_cells + 1)) ||
        (rc = T.get(&d_list_off, (size_t)n_cells + 1)) || (rc = T.get(&d_cell_key, (size_t)n_cells + 1)) || (rc = T.get(&d_max, 1)))
        return rc;
    hipLaunchKernelGGL(cell_records_kernel, dim3(grid_of(n_inc)), dim3(256), 0, st, d_keys_s, n_inc, qbits, d_cflag, d_cidx, d_bflag, d_bidx, d_cell_first,
                       d_cell_key, d_cell_brick);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)d_max, (size_t)1, 0);
    uint32_t *d_rank = NULL, *d_kept = NULL; float* d_near = NULL; unsigned long long* d_total = NULL;
    if (prune) {
        if ((rc = T.get(&d_rank, n_inc + 1)) || (rc = T.get(&d_kept, (size_t)n_cells + 1)) || (rc = T.get(&d_near, (size_t)n_cells + 1)) || (rc = T.get(&d_total, 1))) return rc;
        // positions the kernels assign to a cell lie inside its box up to the float rounding of floor((q - o) * inv_h): 3 ulp of the
        // offset from the origin; squared distances are evaluated in float with a relative error below 4e-7 each
        double ext = 0; for (int k = 0; k < 3; ++k) ext = std::max(ext, (double)n[k] * h);
        const double hh = 0.5 * h + 2.0e-6 * (ext + 1.0) * 0.5 + 1.0e-7, reach = r + 2.0 * h;
        const double margin = 4.0e-6 * reach * reach;
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)d_total, (size_t)2, 0);
        const int pk = getenv("STOCS_GRID_PRUNE_K") ? atoi(getenv("STOCS_GRID_PRUNE_K")) : 8;   // dominators tried per entry (measurement switch: 4, 8, 16)
#define STOCS_PRUNE_LAUNCH(KV) do { if (short_lists) hipLaunchKernelGGL((prune_kernel<KV, 16>), dim3((unsigned)((n_cells + 15) / 16)), dim3(256), 0, st, G, hh, margin, d_cell_first, d_cell_key, n_cells, (uint32_t)n_inc, d_vals_s, c->d_spos, d_rank, d_kept, d_near); \
                                    else hipLaunchKernelGGL((prune_kernel<KV, 64>), dim3((unsigned)((n_cells + 3) / 4)), dim3(256), 0, st, G, hh, margin, d_cell_first, d_cell_key, n_cells, (uint32_t)n_inc, d_vals_s, c->d_spos, d_rank, d_kept, d_near); } while (0)
        const bool short_lists = (double)n_inc <= 12.0 * (double)n_cells;     // a handful of points per cell: 16 lanes per cell, four cells per wavefront
        if (pk == 4) STOCS_PRUNE_LAUNCH(4); else if (pk == 16) STOCS_PRUNE_LAUNCH(16); else STOCS_PRUNE_LAUNCH(8);
#undef STOCS_PRUNE_LAUNCH
        hipLaunchKernelGGL(cell_padded_kept_kernel, dim3(grid_of(n_cells)), dim3(256), 0, st, d_kept, n_cells, 8u, d_padded, d_max, d_total);
    } else
    hipLaunchKernelGGL(cell_padded_kernel, dim3(grid_of(n_cells)), dim3(256), 0, st, d_cell_first, n_cells, (uint32_t)n_inc, 8u, d_padded, d_max);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, st, (int32_t*)(d_padded + n_cells), (size_t)1, 0);
    STOCS_HIP_CHECK(exclusive_scan(d_t32, t32, d_padded, d_list_off, (size_t)n_cells + 1, st));
    uint32_t* tail = rb32 + 3;
    STOCS_HIP_CHECK(hipMemcpyAsync(&tail[0], d_list_off + n_cells, 4, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(&tail[1], d_max, 4, hipMemcpyDeviceToHost, st));
    if (prune) STOCS_HIP_CHECK(hipMemcpyAsync(&rb64[1], d_total, 8, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    const unsigned long long n_kept = prune ? rb64[1] : (unsigned long long)n_inc;
    const size_t n_list = tail[0];
    if (tail[1] > 65535u) { set_error("more than 65535 scene points within epsilon of one grid cell"); return STOCS_ERR_INVALID; }
    // ---- 4. cell words, top table, lists ----
    if ((rc = c->grid_mem.take(std::max<size_t>((size_t)n_bricks * 512, 1) * sizeof(uint4), (void**)&g.d_cells)) ||
        (rc = c->grid_mem.take(std::max<size_t>(n_list, 8) * sizeof(float4), (void**)&g.d_list)))
        return rc;
    hipLaunchKernelGGL(zero_cells_kernel, dim3(grid_of((size_t)n_bricks * 512)), dim3(256), 0, st, g.d_cells, (size_t)n_bricks * 512);
    {   // flat copy of the cell words for the sparse-scene kernel: one look-up per query instead of two dependent ones
        const size_t n_flat = (size_t)n[0] * n[1] * n[2];
        if (!dense && div == 1 && c->lcp_flat && n_flat * sizeof(uint4) <= ((size_t)512 << 20)) {
            if ((rc = c->grid_mem.take(n_flat * sizeof(uint4), (void**)&g.d_flat))) return rc;
            hipLaunchKernelGGL(zero_cells_kernel, dim3(grid_of(n_flat)), dim3(256), 0, st, g.d_flat, n_flat);
        }
    }
    hipLaunchKernelGGL(fill_list_kernel, dim3(grid_of(std::max<size_t>(n_list, 8))), dim3(256), 0, st, g.d_list, std::max<size_t>(n_list, 8));
    hipLaunchKernelGGL(cell_words_kernel, dim3((unsigned)((n_cells + 3) / 4)), dim3(256), 0, st, G, div, d_cell_first, d_cell_key, d_cell_brick, d_list_off, n_cells,
                       (uint32_t)n_inc, d_vals_s, c->d_spos, g.d_top, g.d_cells, g.d_flat, (const uint32_t*)d_kept);
    hipLaunchKernelGGL(list_fill_kernel, dim3(grid_of(n_inc)), dim3(256), 0, st, d_cflag, d_cidx, d_cell_first, d_list_off, d_vals_s, n_inc, c->d_spos, g.d_list, (const uint32_t*)d_rank);
    STOCS_HIP_CHECK(hipGetLastError());
    if (prune && div > 1) {   // no sub-cell masks on grids finer than epsilon: the z word takes the distance bound (has_nearest)
        hipLaunchKernelGGL(cell_nearest_store_kernel, dim3(grid_of(n_cells)), dim3(256), 0, st, d_cell_key, d_cell_brick, n_cells, (const float*)d_near, g.d_cells);
        g.has_nearest = true;
    }
    // ---- 5. dense scenes: chunk bounds ----
    if (dense) {
        if ((rc = c->grid_mem.take(std::max<size_t>(n_list / 8, 1) * sizeof(float), (void**)&g.d_chunk_r))) return rc;
        hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_of(n_list / 8)), dim3(256), 0, st, (int32_t*)g.d_chunk_r, n_list / 8, 0);
        hipLaunchKernelGGL(chunk_bounds_kernel, dim3(grid_of(n_cells)), dim3(256), 0, st, G, d_cell_first, d_cell_key, d_list_off, n_cells, (uint32_t)n_inc,
                           g.d_list, g.d_chunk_r);
        g.has_nearest = div > 1;
        if (g.has_nearest)
            hipLaunchKernelGGL(cell_nearest_kernel, dim3(grid_of(n_cells)), dim3(256), 0, st, d_cell_key, d_cell_brick, d_list_off, n_cells, g.d_chunk_r, g.d_cells);
        STOCS_HIP_CHECK(hipGetLastError());
    }
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    g.n_bricks = (int)n_bricks;
    g.n_entries = (int64_t)n_list;
    g.avg_list_len = n_cells ? (double)n_kept / (double)n_cells : 0.0;
    g.avg_dilated_len = n_cells ? (double)n_inc / (double)n_cells : 0.0;
    g.pruned = prune != 0;
    return STOCS_OK;
}

// ---------------------------------------------------------------------------------------------
// Distance field for the patch test of the scan kernels (SceneGrid::d_dist).  A 64-point step of the model whose bounding
// sphere, under a candidate transform, is farther than epsilon from every scene point cannot contribute to the score
// (stocs.cpp:1019-1024: a model point only counts with a scene point within epsilon), so the kernel skips it after ONE
// look-up here instead of 64 in the cell table.  Per coarse cell: the distance from the cell's centre to the nearest scene
// point (float, shortened by more than its rounding), capped; a position x in the cell is at least value - |x - centre| from the scene.
// ---------------------------------------------------------------------------------------------
struct CullGeom {
    double o[3], g, cap;
    int n[3], w;
};

__global__ __launch_bounds__(256) void dist_fill_kernel(float* __restrict__ t, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = v;
}

// One wavefront per scene point: 16 lanes walk an x row of the point's window, the four quarter-waves take four (y, z) rows at a
// time, so that an access touches 4 table lines instead of 64 and a point's set-up is done once.  Float arithmetic with the cell
// centre computed exactly as the scan kernel computes it; the stored value is shortened by more than every rounding on the way.
// Distances are non-negative floats, whose bit patterns order like unsigned integers: an integer atomic minimum, tried only when a
// device-scope read is larger (a plain load may be served by this XCD's L2 with a value other XCDs have lowered long ago).
// 0.23 ms at Cm (20 000 points x 2 197 cells: 44 M visits, 23 M inside the cap) -- and that is the memory-side atomic minima on
// lines that ~20 points fight over at the same time, not the shape of the loop: a thread per (point, row) walking x alone took
// 0.46 ms (64 lines per access), a workgroup per 16 points and row 0.24, this form 0.23, four reads in flight per lane 0.31.
__global__ __launch_bounds__(256) void dist_splat_kernel(CullGeom G, float ox, float oy, float oz, float g, float cap, const float4* __restrict__ spos, int nS,
                                                         uint32_t* __restrict__ t) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nS) return;
    const int lane = threadIdx.x & 63, xo0 = lane & 15, rsub = lane >> 4;
    const int side = 2 * G.w + 1;
    const float4 p = spos[i];
    const float inv_g = 1.0f / g, cap2 = cap * cap;
    const int ix = (int)floorf((p.x - ox) * inv_g), iy0 = (int)floorf((p.y - oy) * inv_g) - G.w, iz0 = (int)floorf((p.z - oz) * inv_g) - G.w;
    for (int r = rsub; r < side * side; r += 4) {
        const int iz = iz0 + r / side, iy = iy0 + r % side;
        if (iy < 0 || iy >= G.n[1] || iz < 0 || iz >= G.n[2]) continue;
        const float dy = p.y - (oy + ((float)iy + 0.5f) * g), dz = p.z - (oz + ((float)iz + 0.5f) * g);
        const float dyz = dy * dy + dz * dz;
        if (dyz >= cap2) continue;
        const size_t row = ((size_t)iz * G.n[1] + iy) * G.n[0];
        for (int xo = xo0; xo < side; xo += 16) {
            const int cx = ix - G.w + xo;
            if (cx < 0 || cx >= G.n[0]) continue;
            const float dx = p.x - (ox + ((float)cx + 0.5f) * g);
            const float d2 = dx * dx + dyz;
            if (d2 >= cap2) continue;
            const float v = fmaxf(sqrtf(d2) * (1.0f - 2.0e-6f) - 1.0e-6f, 0.0f);
            const uint32_t u = __float_as_uint(v);
            if (u < __hip_atomic_load(&t[row + cx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&t[row + cx], u);
        }
    }
}

// Geometry and memory of the field for the current grid, scene box and model (the cap follows the model's patch radii: a
// patch can only be ruled out when field value - half a cell diagonal exceeds its radius + epsilon).  Called at the end of a
// grid build so that filling it later allocates nothing.
int prepare_cull_field(stocs_ctx* c) {
    SceneGrid& g = c->grid;
    g.d_dist = NULL; g.dist_ready = false;
    if (c->nS <= 0 || c->nM < 64 || !c->d_mpatch) return STOCS_OK;
    const double eps = (double)c->prm.distance_threshold;
    if (!(eps > 0)) return STOCS_OK;
    double cg = eps;
    const double cap = std::min((double)c->patch_r_ref, 8.0 * eps) + eps + cg;
    double ext[3];
    int64_t cells = 0;
    for (;;) {   // coarser cells for boxes that would need more than 16 M of them
        cells = 1;
        for (int k = 0; k < 3; ++k) { ext[k] = (g.bb_mx[k] - g.bb_mn[k]) + 2.0 * (cap + cg); cells *= (int64_t)floor(ext[k] / cg) + 1; }
        if (cells <= ((int64_t)16 << 20)) break;
        cg *= 1.25;
    }
    g.cg_g = (float)cg; g.cg_inv_g = (float)(1.0 / (double)g.cg_g);
    g.cg_cap = (float)cap;
    g.cg_ox = (float)(g.bb_mn[0] - cap - cg); g.cg_oy = (float)(g.bb_mn[1] - cap - cg); g.cg_oz = (float)(g.bb_mn[2] - cap - cg);
    const double of[3] = {g.cg_ox, g.cg_oy, g.cg_oz};
    int n[3];
    for (int k = 0; k < 3; ++k) n[k] = (int)floor((g.bb_mx[k] + cap + cg - of[k]) / (double)g.cg_g) + 1;
    g.cg_nx = n[0]; g.cg_ny = n[1]; g.cg_nz = n[2];
    g.cg_w = (int)ceil(cap / (double)g.cg_g);
    return c->grid_mem.take((size_t)n[0] * n[1] * n[2] * sizeof(float), (void**)&g.d_dist);
}

int fill_cull_field(stocs_ctx* c, hipStream_t st) {
    SceneGrid& g = c->grid;
    if (!g.d_dist || g.dist_ready) return STOCS_OK;
    if (!st) st = c->stream;
    CullGeom G;
    G.o[0] = g.cg_ox; G.o[1] = g.cg_oy; G.o[2] = g.cg_oz;   // the float origin and edge the scan kernels use, widened
    G.g = g.cg_g; G.cap = g.cg_cap;
    G.n[0] = g.cg_nx; G.n[1] = g.cg_ny; G.n[2] = g.cg_nz; G.w = g.cg_w;
    const size_t n = (size_t)g.cg_nx * g.cg_ny * g.cg_nz;
    hipLaunchKernelGGL(dist_fill_kernel, dim3(grid_of(n)), dim3(256), 0, st, g.d_dist, n, g.cg_cap);
    hipLaunchKernelGGL(dist_splat_kernel, dim3((unsigned)((c->nS + 3) / 4)), dim3(256), 0, st, G, g.cg_ox, g.cg_oy, g.cg_oz, g.cg_g, g.cg_cap,
                       c->d_spos, c->nS, (uint32_t*)g.d_dist);
    STOCS_HIP_CHECK(hipGetLastError());
    g.dist_ready = true;
    return STOCS_OK;
}

}  // namespace stocs
