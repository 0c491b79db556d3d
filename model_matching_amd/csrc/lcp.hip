// lcp.hip -- batched weighted-LCP verification, THE METRIC KERNEL ("candidate poses verified / s").
// Replaces stocs_estimator::compute_alignment_score_for_rigid_transform (reference
// src/stocs.cpp:1006-1041) and the kd-tree query it calls per model point
// (reference include/super4pcs/accelerators/kdtree.h:394-459).
//
// Mapping: one candidate transform per workgroup of four wavefronts, which take the candidate's 64-point steps in turn and
// add their integer partial sums through LDS at the end (the only barrier); small models run one wavefront per candidate.
// Several candidates per workgroup, and a model tile shared through LDS, were measured and lost.
// The 3x4 transform is wave-uniform (scalar loads -> SGPRs).  A lane takes one patch-ordered model
// point per step: coalesced 16-byte loads, and the 64 queries of one step fall into a handful
// of neighbouring grid cells, so the cell/list gathers of a wave share cache lines.
// Per query: transform the point, locate its cell (one word of the flat cell table, or top -> brick -> cell word), scan that cell's
// candidate list (every scene point within epsilon of the cell box) for the nearest point with
// d^2 <= epsilon^2, then the 30-degree normal test as an exact threshold on the dot product, and an
// integer (2^32 fixed-point) sum of the class-probability weights.  No atomics: results are run-to-run
// deterministic and do not depend on the batch or on how the points are split over wavefronts.
// The queue-fed kernel (lcp_coopq_kernel, the automatic choice) first rules out whole 64-point steps whose bounding
// sphere is out of reach of the scene (patch test, one look-up in a distance field of the scene), collects the
// queries that survive the sub-cell mask in a per-wave LDS ring and verifies them 16 at a time, four lanes per query
// with two entries of a 128-byte list line each.
//
// Roofline: HBM-read model, algorithmic bytes 68 + 52*|M| per pose.
#include <stdlib.h>
#include <string.h>

#include <cstring>


#include <algorithm>

#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

struct LcpArgs {
    const float4* mpos;   // centred model positions in patch order (ctx.hip), NaN-padded to whole 64-point steps + one
    const float4* mnrm;
    const int32_t* mperm; // sorted slot -> original model index (detail output only)
    int M;
    const int32_t* top;
    const uint4* cells;
    const uint4* flat;    // cell words addressed by (cz*ny + cy)*nx + cx, or NULL (brick look-up through top / cells)
    const float4* list;
    const float4* snrmw;  // scene unit normal + class-probability weight
    // instance-mode trial batches (stocs_run_trials): candidate i belongs to trial cand_trial[i] of the batch and adds the class
    // probabilities as ITS trial's sampling decayed them (Q8): that trial's copy lies cand_trial[i] * snrmw_stride bytes behind snrmw
    const int32_t* cand_trial;   // NULL: every candidate scores against snrmw itself
    size_t snrmw_stride;
    const float* chunk_r; // per 8-entry chunk: lower bound of |entry - cell centre| (dense scenes), else NULL
    float ox, oy, oz, inv_h, inv_h4, h;
    int nx, ny, nz, nbx, nby;
    float sq_eps, dot_lo, eps;
    float bound_margin;     // slack of the early-exit / nearest-point bound tests of the centre-sorted lists: relative to epsilon and to the
                            // scene's coordinate magnitude (the float rounding of |query - cell centre| scales with the coordinates)
    int has_nearest;        // the z word of a dense grid's cell is a lower bound of |cell centre - nearest listed point| (SceneGrid::has_nearest)
    const int32_t* order;   // processing slot -> candidate (NULL: identity): candidates that land in the same part of the scene run together
    int xcd_blocks;         // != 0: workgroups of one XCD take a contiguous run of slots (each XCD has its own L2)
    unsigned long long* best;   // != NULL: compute_best_transform in the kernel's epilogue -- every candidate's packed (score, ~id) key joins an
    uint32_t id_offset;         // atomic max on this word (integer max: order independent), id = id_offset + candidate
    // patch test (lcp_coopq_kernel): bounding sphere per 64-point step + the scene's distance field (SceneGrid::d_dist); patch == NULL: off
    const float4* patch;
    const float* dist;
    float gox, goy, goz, g, inv_g, cap;
    int gnx, gny, gnz;
#ifdef STOCS_TOOLS_BUILD
    int ablate;             // measurement build only (STOCS_LCP_ABLATE): parts of the kernel switched off to price them; scores are then wrong
#endif
};
#ifdef STOCS_TOOLS_BUILD
#define STOCS_ABLATE(a, bit) (((a).ablate & (bit)) != 0)
#else
#define STOCS_ABLATE(a, bit) false
#endif

// workgroup -> first processing slot.  The hardware hands consecutive workgroups to the 8 XCDs round-robin; with
// xcd_blocks the workgroups that land on one XCD take consecutive slot blocks, so an XCD's L2 sees one
// contiguous part of the (spatially ordered) candidate list.  Then slot -> candidate through the order array.
__device__ __forceinline__ int lcp_candidate(const LcpArgs& a, int n, int w, int wpb = 4) {
    int blk = blockIdx.x;
    if (a.xcd_blocks == 1) {
        const int nb = gridDim.x, per = nb >> 3, rem = nb & 7, xcd = blk & 7;
        blk = xcd * per + (xcd < rem ? xcd : rem) + (blk >> 3);
    } else if (a.xcd_blocks > 1) {   // chunks of xcd_blocks consecutive blocks per XCD, chunks dealt round-robin (balanced)
        const int C = a.xcd_blocks, nb = gridDim.x, full = (nb / (8 * C)) * (8 * C);
        if (blk < full) {
            const int xcd = blk & 7, j = blk >> 3;
            blk = ((j / C) * 8 + xcd) * C + (j % C);
        }
    }
    const int slot = __builtin_amdgcn_readfirstlane(blk * wpb + w);
    if (slot >= n) return -1;
    return a.order ? __builtin_amdgcn_readfirstlane(a.order[slot]) : slot;
}


// the scene normals + weights candidate `cand` scores against (wave-uniform: scalar loads)
__device__ __forceinline__ const float4* lcp_weights_of(const LcpArgs& a, int cand) {
    if (!a.cand_trial) return a.snrmw;
    const int t = __builtin_amdgcn_readfirstlane(a.cand_trial[cand < 0 ? 0 : cand]);
    return (const float4*)((const char*)a.snrmw + (size_t)t * a.snrmw_stride);
}

#define DPP_QUAD_XOR1 0xB1   /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E   /* quad_perm [2,3,0,1] */
#define DPP_HALF_MIRROR 0x141 /* lane k <-> 7-k inside each group of 8 */
#define DPP_ROW_SHR(n) (0x110 + (n))

// Score accumulation.  The reference adds class probabilities in a sequential float loop (stocs.cpp:1033); a parallel
// kernel cannot keep that order, so the weights are added as 2^32 fixed-point integers instead: exact for every weight
// >= 2^-9 (class probabilities are >= the 0.10 threshold), hence independent of the order -- of the lane a point lands on,
// of how a candidate's model points are split over wavefronts, of the batch it is scored in.  The score is the correctly
// rounded float of the exact mean; the reference's running float sum differs from it by ~1e-7 (tests: <= 1e-5).
__device__ __forceinline__ void lcp_add(unsigned long long& acc, float w) {
    if (w > 0.0f) acc += (unsigned long long)(w * 4294967296.0f);   // exact product (power of two), truncating conversion
}
__device__ __forceinline__ unsigned long long lcp_wave_sum(unsigned long long acc) {
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return acc;
}
__device__ __forceinline__ float lcp_finish(unsigned long long total, int M) {
    return (float)((double)total * (1.0 / 4294967296.0) / (double)M);
}
// the candidate's score, and -- when the caller wants the arg-max with it -- its key of compute_best_transform (stocs.cpp:987-998:
// larger score wins, lower id wins ties, scores that are not positive never win): one atomic per candidate, no second kernel
__device__ __forceinline__ void lcp_store(const LcpArgs& a, float* __restrict__ out, int cand, unsigned long long total) {
    const float s = lcp_finish(total, a.M);
    out[cand] = s;
    if (a.best && s > 0.0f)
        atomicMax(a.best, ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (a.id_offset + (uint32_t)cand)));
}

// candidate update: inclusive radius (kdtree.h:424), ties -> larger scene index.  When a list is stored in
// ascending index order (IDX: every grid except the dense, centre-sorted one) `<=` implements the tie rule.
template <bool IDX>
__device__ __forceinline__ void take_if_better(float d, int ei, float& gd, int& gi) {
    if (IDX) { if (d <= gd) { gd = d; gi = ei; } }
    else { if (d < gd || (d == gd && ei > gi)) { gd = d; gi = ei; } }
}

template <bool DETAIL, bool IDX = true>
__global__ __launch_bounds__(256) void lcp_kernel(LcpArgs a, const float* __restrict__ T16, float* __restrict__ out,
                                                  int n, int32_t* __restrict__ hit_out, uint8_t* __restrict__ cnt_out) {
    const int lane = threadIdx.x & 63;
    const int cand = lcp_candidate(a, n, threadIdx.x >> 6);
    if (cand < 0) return;
    const float* T = T16 + (size_t)cand * 16;
    const float4* __restrict__ snrmw = lcp_weights_of(a, cand);
    const float t0 = T[0], t1 = T[1], t2 = T[2], t4 = T[4], t5 = T[5], t6 = T[6], t8 = T[8], t9 = T[9], t10 = T[10],
                t12 = T[12], t13 = T[13], t14 = T[14];
    unsigned long long acc = 0ull;
    for (int i = lane; i < a.M; i += 64) {
        const float4 p = a.mpos[i];
        // (mat * p.homogeneous()).head<3>()
        const float qx = ((t0 * p.x + t4 * p.y) + t8 * p.z) + t12;
        const float qy = ((t1 * p.x + t5 * p.y) + t9 * p.z) + t13;
        const float qz = ((t2 * p.x + t6 * p.y) + t10 * p.z) + t14;
        const float fx = floorf((qx - a.ox) * a.inv_h);
        const float fy = floorf((qy - a.oy) * a.inv_h);
        const float fz = floorf((qz - a.oz) * a.inv_h);
        int best = -1;
        if (fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)a.nx && fy < (float)a.ny && fz < (float)a.nz) {
            const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
            const int brick = a.top[((cz >> 3) * a.nby + (cy >> 3)) * a.nbx + (cx >> 3)];
            if (brick >= 0) {
                const uint4 cw = a.cells[(size_t)brick * 512 + (((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7))];
                float bd = a.sq_eps;
                for (uint32_t k = 0; k < cw.y; ++k) {
                    const float4 s = a.list[cw.x + k];
                    const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
                    const float d = dx * dx + (dy * dy + dz * dz);
                    take_if_better<IDX>(d, __float_as_int(s.w), bd, best);
                }
            }
        }
        bool counted = false;
        if (best >= 0) {
            const float4 nm = a.mnrm[i];
            // mat.block<3,3>(0,0) * normal
            const float nx = t0 * nm.x + (t4 * nm.y + t8 * nm.z);
            const float ny = t1 * nm.x + (t5 * nm.y + t9 * nm.z);
            const float nz = t2 * nm.x + (t6 * nm.y + t10 * nm.z);
            const float4 sn = snrmw[best];
            const float d = sn.x * nx + (sn.y * ny + sn.z * nz);
            // acos(d)*180/pi < 30 as an exact threshold; d > 1 -> NaN angle -> not counted (Q7)
            counted = (d >= a.dot_lo) && (d <= 1.0f);
            if (counted) lcp_add(acc, sn.w);
        }
        if (DETAIL) {
            const int orig = a.mperm[i];
            hit_out[(size_t)cand * a.M + orig] = best;
            cnt_out[(size_t)cand * a.M + orig] = counted ? 1 : 0;
        }
    }
    // fixed-shape butterfly: deterministic
    acc = lcp_wave_sum(acc);
    if (lane == 0) lcp_store(a, out, cand, acc);
}

// ---------------------------------------------------------------------------------------------
// Variant 1 ("coop8"): what the ISA of v0 shows (profiles/r01_lcp_analysis.md): its list loop issues
// ~18 instructions per trip for the ~18 of 64 lanes that own a list, i.e. about one issued
// wave-instruction per list entry, and every entry is its own 16-byte gather.  Here the hit queries
// of a step are compacted through LDS and each 8-lane group takes ONE query: the group streams the
// query's 8-padded list one 128-byte line per load (all 64 lanes useful), keeps its running best in
// registers across chunks, and reduces once per query with DPP (pure VALU; __shfl/ds_bpermute would go
// through the LDS pipeline).  ~13 instructions per 64 entries instead of ~18 per ~18.
// Tie rule identical to v0: smallest d^2, then largest scene index; same lane -> point assignment
// and reduction tree, so scores are bitwise equal to v0.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// minimum of a squared distance over the 8 lanes of a query group.  Squared distances are non-negative and never NaN here
// (a NaN distance fails `d <= best` and is never kept), so their bit patterns order like unsigned integers: an integer
// minimum needs no NaN canonicalisation and takes its DPP operand directly -- 3 instructions instead of 10.
template <int GL = 8>
__device__ __forceinline__ float group_min_nonneg(float v) {
    uint32_t u = __float_as_uint(v);
    if (GL >= 2) u = min(u, (uint32_t)__builtin_amdgcn_update_dpp((int)u, (int)u, DPP_QUAD_XOR1, 0xF, 0xF, false));
    if (GL >= 4) u = min(u, (uint32_t)__builtin_amdgcn_update_dpp((int)u, (int)u, DPP_QUAD_XOR2, 0xF, 0xF, false));
    if (GL == 8) u = min(u, (uint32_t)__builtin_amdgcn_update_dpp((int)u, (int)u, DPP_HALF_MIRROR, 0xF, 0xF, false));
    return __uint_as_float(u);
}
__device__ __forceinline__ float group8_min_nonneg(float v) { return group_min_nonneg<8>(v); }

template <bool DETAIL, int UNR, bool MASK = true, bool EARLY = false, bool IDX = true, int WPB = 4, bool SPLIT = false>
__global__ __launch_bounds__(64 * WPB) void lcp_coop_kernel(LcpArgs a, const float* __restrict__ T16, float* __restrict__ out,
                                                       int n, int32_t* __restrict__ hit_out, uint8_t* __restrict__ cnt_out) {
    __shared__ float4 qt[WPB][64];     // per lane: qx, qy, qz, bits(list offset)
    __shared__ uint32_t qn[WPB][64];   // per lane: list length
    __shared__ uint32_t hl[WPB][64];   // compacted hit list: r-th hit lane
    __shared__ float rd[WPB][64];      // per lane: best d^2
    __shared__ int ri[WPB][64];        // per lane: best scene index
    __shared__ float qd[WPB][64];      // per lane: |query - cell centre| (EARLY)
    __shared__ unsigned long long part[WPB];
    const int lane = threadIdx.x & 63;
    const int sub = lane & 7, grp = lane >> 3;
    const int w = threadIdx.x >> 6;
    // SPLIT: the WPB wavefronts of the workgroup share one candidate and take its 64-point steps round-robin (see lcp_coopq_kernel)
    const int cand = SPLIT ? lcp_candidate(a, n, 0, 1) : lcp_candidate(a, n, w, WPB);
    if (cand < 0) return;
    const int first = SPLIT ? 64 * w : 0, stride = SPLIT ? 64 * WPB : 64;
    const float4* __restrict__ snrmw = lcp_weights_of(a, cand);
    const float* T = T16 + (size_t)cand * 16;
    const float t0 = T[0], t1 = T[1], t2 = T[2], t4 = T[4], t5 = T[5], t6 = T[6], t8 = T[8], t9 = T[9], t10 = T[10],
                t12 = T[12], t13 = T[13], t14 = T[14];
    unsigned long long acc = 0ull;
    for (int base = first; base < a.M; base += stride) {
        const int i = base + lane;
        float qx = 0.f, qy = 0.f, qz = 0.f, qcd = 0.f;
        uint32_t off = 0, cnt = 0;
        if (i < a.M) {
            const float4 p = a.mpos[i];
            qx = ((t0 * p.x + t4 * p.y) + t8 * p.z) + t12;
            qy = ((t1 * p.x + t5 * p.y) + t9 * p.z) + t13;
            qz = ((t2 * p.x + t6 * p.y) + t10 * p.z) + t14;
            const float ux = (qx - a.ox) * a.inv_h, uy = (qy - a.oy) * a.inv_h, uz = (qz - a.oz) * a.inv_h;
            const float fx = floorf(ux), fy = floorf(uy), fz = floorf(uz);
            if (fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)a.nx && fy < (float)a.ny && fz < (float)a.nz) {
                const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
                const int brick = STOCS_ABLATE(a, 1) ? -1 : a.top[((cz >> 3) * a.nby + (cy >> 3)) * a.nbx + (cx >> 3)];   // 1: no look-ups at all
                if (brick >= 0) {
                    uint4 cw = make_uint4(0u, 0u, 0u, 0u);
                    if (!STOCS_ABLATE(a, 128)) cw = a.cells[(size_t)brick * 512 + (((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7))];   // 128: top table only
                    // sub-cell filter: no scene point within epsilon of this 1/4-cell => no neighbour possible
                    const int sb = ((int)((uz - fz) * 4.0f) << 4) | ((int)((uy - fy) * 4.0f) << 2) | (int)((ux - fx) * 4.0f);
                    const uint32_t mw = a.has_nearest ? 0xFFFFFFFFu : (sb < 32 ? cw.z : cw.w);   // (has_nearest: no mask on this grid, z holds a distance)
                    off = cw.x; cnt = (!MASK || ((mw >> (sb & 31)) & 1u)) ? cw.y : 0u;
                    if (STOCS_ABLATE(a, 2)) cnt = cw.x == 0xFFFFFFF1u ? 1u : 0u;   // 2: look-ups done, nobody survives
                    if (EARLY) {
                        const float ex = qx - (a.ox + ((float)cx + 0.5f) * a.h), ey = qy - (a.oy + ((float)cy + 0.5f) * a.h),
                                    ez = qz - (a.oz + ((float)cz + 0.5f) * a.h);
                        qcd = sqrtf(ex * ex + (ey * ey + ez * ez));
                    }
                }
            }
        }
        const bool hit = cnt != 0;
        const unsigned long long mask = __ballot(hit);
        int best = -1;
        if (mask) {
            const int nh = __popcll(mask);
            if (hit) {
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                hl[w][rank] = (uint32_t)lane;
                qt[w][lane] = make_float4(qx, qy, qz, __int_as_float((int)off));
                qn[w][lane] = cnt;
                if (EARLY) qd[w][lane] = qcd;
            }
            __builtin_amdgcn_wave_barrier();
            for (int s = 0; s < nh; s += 8) {
                const int slot = s + grp;
                const bool gact = slot < nh;
                const uint32_t L = gact ? hl[w][slot] : 0u;
                const float4 qq = qt[w][L];
                const uint32_t c = gact ? qn[w][L] : 0u;
                const float4* lp = a.list + (uint32_t)__float_as_int(qq.w) + sub;
                float gd = a.sq_eps;
                int gi = -1;
                if (EARLY) {
                    // lists are sorted by distance from the cell centre: stop at the first chunk that the
                    // triangle inequality rules out (|q - p| >= |p - c| - |q - c| > sqrt(best) for all later p)
                    const float qcg = qd[w][L];
                    const uint32_t chunk0 = (uint32_t)__float_as_int(qq.w) >> 3;
                    uint32_t nchunks = (c + 7u) >> 3;
                    float gb = a.sq_eps;   // best d^2 of the whole group so far
                    if (STOCS_ABLATE(a, 4)) nchunks = min(nchunks, (uint32_t)UNR);   // 4: the first trip of every list only
                    for (uint32_t j = 0; __any(j < nchunks); j += UNR) {
                        if (j < nchunks && j > 0 && (STOCS_ABLATE(a, 256) ? 0.0f : a.chunk_r[chunk0 + j]) - qcg > sqrtf(gb) + a.bound_margin) nchunks = 0;   // 256: no chunk bounds (scan everything)
                        float4 e[UNR];
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            e[u] = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
                            if (j + u < nchunks) e[u] = lp[(j + u) << 3];
                        }
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            const float dx = qq.x - e[u].x, dy = qq.y - e[u].y, dz = qq.z - e[u].z;
                            const float d = dx * dx + (dy * dy + dz * dz);
                            const int ei = __float_as_int(e[u].w);
                            if (d < gd || (d == gd && ei > gi)) { gd = d; gi = ei; }   // lists are no longer in index order
                        }
                        gb = fminf(gd, dpp_f32<DPP_QUAD_XOR1>(gd));
                        gb = fminf(gb, dpp_f32<DPP_QUAD_XOR2>(gb));
                        gb = fminf(gb, dpp_f32<DPP_HALF_MIRROR>(gb));
                    }
                } else
                for (uint32_t k = 0; __any(k < c); k += 8 * UNR) {
                    float4 e[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        e[u] = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
                        if (k + 8 * u < c) e[u] = lp[k + 8 * u];   // whole 128-byte line per group; tail entries are sentinels
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const float dx = qq.x - e[u].x, dy = qq.y - e[u].y, dz = qq.z - e[u].z;
                        const float d = dx * dx + (dy * dy + dz * dz);
                        take_if_better<IDX>(d, __float_as_int(e[u].w), gd, gi);
                    }
                }
                // group minimum of d^2 (DPP), then the largest index among the lanes that hold it
                float dm = fminf(gd, dpp_f32<DPP_QUAD_XOR1>(gd));
                dm = fminf(dm, dpp_f32<DPP_QUAD_XOR2>(dm));
                dm = fminf(dm, dpp_f32<DPP_HALF_MIRROR>(dm));
                int im = (gd == dm) ? gi : -1;
                im = max(im, dpp_i32<DPP_QUAD_XOR1>(im));
                im = max(im, dpp_i32<DPP_QUAD_XOR2>(im));
                im = max(im, dpp_i32<DPP_HALF_MIRROR>(im));
                if (gact && sub == 0) { rd[w][L] = dm; ri[w][L] = im; }
            }
            __builtin_amdgcn_wave_barrier();
            if (hit) best = ri[w][lane];
            __builtin_amdgcn_wave_barrier();
        }
        bool counted = false;
        if (best >= 0 && STOCS_ABLATE(a, 8)) lcp_add(acc, 0.5f);   // 8: no normal test
        else if (best >= 0) {
            const float4 nm = a.mnrm[i];
            const float nx = t0 * nm.x + (t4 * nm.y + t8 * nm.z);
            const float ny = t1 * nm.x + (t5 * nm.y + t9 * nm.z);
            const float nz = t2 * nm.x + (t6 * nm.y + t10 * nm.z);
            const float4 sn = snrmw[best];
            const float d = sn.x * nx + (sn.y * ny + sn.z * nz);
            counted = (d >= a.dot_lo) && (d <= 1.0f);
            if (counted) lcp_add(acc, sn.w);
        }
        if (DETAIL && i < a.M) {
            const int orig = a.mperm[i];
            hit_out[(size_t)cand * a.M + orig] = best;
            cnt_out[(size_t)cand * a.M + orig] = counted ? 1 : 0;
        }
    }
    acc = lcp_wave_sum(acc);
    if (SPLIT) {
        if (lane == 0) part[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long total = 0;
#pragma unroll
            for (int k = 0; k < WPB; ++k) total += part[k];
            lcp_store(a, out, cand, total);
        }
    } else if (lane == 0) lcp_store(a, out, cand, acc);
}

// ---------------------------------------------------------------------------------------------
// Patch test.  The 64 model points of a step lie in a sphere (centre c, radius r; ctx.hip).  Under the candidate transform
// x -> A x + t every one of them lands within s * r of A c + t, s >= the largest singular value of A (1 for a rigid
// transform; bounded here by the square root of the largest absolute row sum of A^T A, so that a caller's non-rigid matrix
// is handled too).  The scene's distance field gives a lower bound of the distance from A c + t to the nearest scene point;
// when that exceeds s * r + epsilon no point of the step has a scene point within epsilon, the step adds nothing to the
// score (stocs.cpp:1019-1024) and is skipped: one look-up per step instead of 64.  Margins cover the float rounding of the
// transform, of the field and of the cell centre; positions that are not finite are never ruled out here.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lcp_linear_norm_bound(float t0, float t1, float t2, float t4, float t5, float t6, float t8, float t9, float t10) {
    const float g00 = t0 * t0 + (t1 * t1 + t2 * t2), g11 = t4 * t4 + (t5 * t5 + t6 * t6), g22 = t8 * t8 + (t9 * t9 + t10 * t10);
    const float g01 = fabsf(t0 * t4 + (t1 * t5 + t2 * t6)), g02 = fabsf(t0 * t8 + (t1 * t9 + t2 * t10)), g12 = fabsf(t4 * t8 + (t5 * t9 + t6 * t10));
    const float rho = fmaxf(g00 + (g01 + g02), fmaxf(g11 + (g01 + g12), g22 + (g02 + g12)));
    return sqrtf(rho) * 1.00001f;   // NaN / inf entries give NaN / inf: nothing is ruled out then (every comparison below fails)
}
__device__ __forceinline__ bool lcp_patch_dead(const LcpArgs& a, const float4 sp, float sn, float t0, float t1, float t2, float t4, float t5, float t6,
                                               float t8, float t9, float t10, float t12, float t13, float t14) {
    const float cx = ((t0 * sp.x + t4 * sp.y) + t8 * sp.z) + t12;
    const float cy = ((t1 * sp.x + t5 * sp.y) + t9 * sp.z) + t13;
    const float cz = ((t2 * sp.x + t6 * sp.y) + t10 * sp.z) + t14;
    const float mag = fabsf(cx) + (fabsf(cy) + fabsf(cz));
    if (!(mag < 1.0e18f)) return false;
    const float need = (sn * sp.w + a.eps * 1.002f) + (4.0e-6f + 4.0e-6f * mag);
    const float ux = (cx - a.gox) * a.inv_g, uy = (cy - a.goy) * a.inv_g, uz = (cz - a.goz) * a.inv_g;
    const float fx = floorf(ux), fy = floorf(uy), fz = floorf(uz);
    // one cell inside the border: the float cell of a position next to the box never names a cell outside the table
    if (fx >= 1.0f && fy >= 1.0f && fz >= 1.0f && fx < (float)(a.gnx - 1) && fy < (float)(a.gny - 1) && fz < (float)(a.gnz - 1)) {
        const float v = a.dist[((uint32_t)(int)fz * (uint32_t)a.gny + (uint32_t)(int)fy) * (uint32_t)a.gnx + (uint32_t)(int)fx];   // <= 16 M cells
        const float ex = cx - (a.gox + (fx + 0.5f) * a.g), ey = cy - (a.goy + (fy + 0.5f) * a.g), ez = cz - (a.goz + (fz + 0.5f) * a.g);
        return v - sqrtf(ex * ex + (ey * ey + ez * ez)) > need;
    }
    // outside (or in the border cells of) the table: at least cap from every scene point
    return a.cap > need;
}

// ---------------------------------------------------------------------------------------------
// Variant 20 ("coop8 + queue"): the cooperative scan of variant 1 fed from a per-wave LDS ring that
// collects the hit queries of successive steps (ballot + mbcnt compaction), so that every 8-lane group
// always owns a query (with ~11 hits per 64-point step the groups of variant 1 are ~69 % busy) and the
// normal test runs on full wavefronts.  Lane <-> point assignment of the accumulation differs from v0,
// so scores agree with v0 to rounding (1e-7), not bitwise; still run-to-run deterministic.
// ---------------------------------------------------------------------------------------------
// SPLIT: the WPB wavefronts of a workgroup share ONE candidate and take its 64-point steps round-robin (a trial's ~8 000
// candidates are a single round of wavefronts on the chip, and four times as many, four times shorter wavefronts finish
// it sooner; big batches lose nothing); the partial sums are integers, so the score is the same bit for bit.
// TILE (tools build, A/B only): the WPB candidates of a workgroup read the model points from a 256-point tile staged in LDS
// (one global load per point and workgroup instead of one per wavefront), at the price of a workgroup barrier per tile.
// EARLY (dense scenes: lists sorted by distance from the cell centre, a lower bound of that distance per 8-entry line): a query
// stops at the first line the triangle inequality rules out, as in variant 31 -- but fed from the queue, so that the first
// lines of 32 queries are in flight together where variant 31 has the two lines of 8.
// NEAR (pruned lists on grids finer than epsilon, round 5): the cell word's z is a lower bound of |cell centre - nearest listed point|
// (no sub-cell mask there); a query farther from the centre than that bound + epsilon never touches the list -- EARLY's first test
// without its line-by-line exit, which lists of ~5 entries have no use for.
// TINY (round 5, with GL = 4 on index-ordered lists): after the dominance pruning most surviving queries see a list of at most four entries.
// Those are verified by TWO lanes each (two entries per lane, 32 queries per trip of the wavefront) instead of four: the texture addresser
// -- the unit this kernel sits on -- takes four lane addresses per clock whatever their width, so a four-entry list read by four lanes x two
// entries costs twice the addresses it needs.  The pending queries are ordered tiny lists first; a trip is a two-lane trip while 32 tiny
// queries are left.
template <bool DETAIL, int UNR, bool SORTQ = false, int PIPE = 4, bool IDX = true, int WPB = 4, int FLAT = 0, bool SPLIT = false, bool TILE = false, bool EARLY = false, int GL = 8, int FIRST = 1, bool NOSENT = false, bool NEAR = false, bool TINY = false>
__global__ __launch_bounds__(64 * WPB, 8) void lcp_coopq_kernel(LcpArgs a, const float* __restrict__ T16, float* __restrict__ out,
                                                        int n, int32_t* __restrict__ hit_out, uint8_t* __restrict__ cnt_out) {
    __shared__ float4 qt[WPB][128];     // qx, qy, qz, bits(list offset)
    __shared__ float qcd[EARLY ? WPB : 1][EARLY ? 128 : 1];   // |query - cell centre| (EARLY)
    __shared__ uint32_t qn[WPB][128];   // list length
    __shared__ uint32_t qs[WPB][128];   // model slot (patch order)
    __shared__ int ri[WPB][128];        // best scene index
    __shared__ uint8_t ord[WPB][64];    // batch order sorted by list length (SORTQ)
    __shared__ unsigned long long part[WPB];
    const int lane = threadIdx.x & 63;
    // GL lanes verify one query together: each reads EPL = 8 / GL entries of a 128-byte list line (GL = 8: one entry per lane, eight
    // queries per trip; GL = 4: two entries per lane, sixteen queries per trip and one reduction level less)
    // FIRST: list lines requested in a query's first trip (the second line of a list that has one arrives with the first: half of
    // the survivors' lists at Cm are longer than a line); not with EARLY, whose later lines wait for the bound test
    constexpr int EPL = 8 / GL, E0 = EPL * FIRST;
    static_assert(!EARLY || FIRST == 1, "early exit decides line by line");
    static_assert(!TINY || (GL == 4 && SORTQ && IDX && !EARLY && PIPE == 1), "two-lane trips: the four-lane queue form on index-ordered lists");
#if defined(STOCS_TOOLS_BUILD) && defined(STOCS_LCP_W_UNIFORM)
    // measurement build only (profiles/r04_lcp_w_uniform_isa.md): the wavefront index forced into an SGPR.  63 VGPRs, 8 waves -- and
    // 12 % SLOWER: every LDS address of the per-wavefront queue (qt[w][..], qn[w][..], ri[w][..]) is then formed as "scalar base +
    // vector index" with the scalar part re-materialised into a VGPR at each use instead of living in one address VGPR
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    const int w = threadIdx.x >> 6;
#endif
    const int cand = SPLIT ? lcp_candidate(a, n, 0, 1) : lcp_candidate(a, n, w, WPB);
    if (cand < 0 && !TILE) return;   // SPLIT: the whole workgroup leaves together; TILE: the wavefront stays for the barriers
    const float4* __restrict__ snrmw = lcp_weights_of(a, cand);
    const float* T = T16 + (size_t)(cand < 0 ? 0 : cand) * 16;
    const float t0 = T[0], t1 = T[1], t2 = T[2], t4 = T[4], t5 = T[5], t6 = T[6], t8 = T[8], t9 = T[9], t10 = T[10],
                t12 = T[12], t13 = T[13], t14 = T[14];
    unsigned long long acc = 0ull;
    int head = 0, tail = 0;

    auto process = [&](int nq) {
        int n_tiny = 0;
        // order the pending queries by list length (number of 128-byte chunks) so that the eight
        // groups of a step stream lists of similar length: a counting sort on <= 8 classes with ballots
        if (SORTQ) {
            // (most lists are one or two lines: three classes -- one line, two, more -- keep the eight groups of a step on
            //  lists of similar length at a third of the instructions of a sort over all lengths)
            uint32_t cls = 4;
            if (lane < nq) {
                const uint32_t cq = qn[w][(head + lane) & 127], nch = (cq + 7u) >> 3;
                cls = (TINY && cq <= 4u) ? 0u : (nch <= (uint32_t)FIRST ? 1u : (nch <= (uint32_t)(FIRST + UNR) ? 2u : 3u));   // at most four entries; one trip, two, more
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long mt = __ballot(cls == 0u), m0 = __ballot(cls == 1u), m1 = __ballot(cls == 2u), m2 = __ballot(cls == 3u);
            const int nt = __popcll(mt), n0 = __popcll(m0), n1 = __popcll(m1);
            n_tiny = nt;
            const int pos = cls == 0u ? __popcll(mt & below) : (cls == 1u ? nt + __popcll(m0 & below) : (cls == 2u ? nt + n0 + __popcll(m1 & below) : nt + n0 + n1 + __popcll(m2 & below)));
            if (lane < nq) ord[w][pos] = (uint8_t)lane;
            __builtin_amdgcn_wave_barrier();
        }
        // PIPE query-steps are in flight together: their first 128-byte list lines (most lists are a single
        // line) are requested back to back, so one memory round trip serves PIPE*8 queries
        for (int s0 = 0; s0 < nq;) {
            const bool two = TINY && s0 + 32 <= n_tiny;          // (uniform) a trip of 32 tiny lists, two lanes each
            const int gl = two ? 2 : GL, NG = two ? 32 : 64 / GL;
            const int sub = lane & (gl - 1), grp = two ? (lane >> 1) : lane / GL;
            float4 e0[PIPE][E0];
            bool wide[PIPE];
            int idxs[PIPE];
            uint32_t cs[PIPE];
#pragma unroll
            for (int u = 0; u < PIPE; ++u) {
                const int slot = s0 + NG * u + grp;
                const bool gact = slot < nq;
                idxs[u] = (head + (SORTQ ? (int)ord[w][gact ? slot : 0] : slot)) & 127;
                cs[u] = gact ? qn[w][idxs[u]] : 0u;
                wide[u] = true;
                if (!NOSENT) {
#pragma unroll
                    for (int e = 0; e < E0; ++e) e0[u][e] = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
                }
                if (STOCS_ABLATE(a, 4)) {   // 4: no list loads (every survivor "hits" scene point 0 at distance 0)
                    if (cs[u]) { const float4 qq0 = qt[w][idxs[u]]; e0[u][0] = make_float4(qq0.x, qq0.y, qq0.z, __int_as_float(0)); }
                    cs[u] = cs[u] ? 1u : 0u;
                }
                else if (NOSENT) {
                    // no sentinels, no predication: a group without a query reads list line 0 (its result is never stored), a list
                    // shorter than the trip reads its last line again -- the same entries, the same winner
                    const float4* l0 = a.list + (cs[u] ? (uint32_t)__float_as_int(qt[w][idxs[u]].w) : 0u) + sub;
                    const uint32_t last = cs[u] ? ((cs[u] - 1u) & ~7u) : 0u;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) e0[u][e] = l0[gl * e];
                    // the later lines of the first trip only when some list of this trip has them (the queries are ordered by list
                    // length, so a third of the trips are single-line lists throughout: no addresses spent on re-reading those)
                    wide[u] = FIRST > 1 && __any(cs[u] > 8u);
                    if (wide[u]) {
#pragma unroll
                        for (int e = EPL; e < E0; ++e) e0[u][e] = l0[min((uint32_t)(8 * (e / EPL)), last) + gl * (e % EPL)];
                    }
                }
                else if (cs[u]) {
                    const float4* l0 = a.list + (uint32_t)__float_as_int(qt[w][idxs[u]].w) + sub;
#pragma unroll
                    for (int e = 0; e < E0; ++e)
                        if (e < EPL || (uint32_t)(8 * (e / EPL)) < cs[u]) e0[u][e] = l0[8 * (e / EPL) + gl * (e % EPL)];
                }
            }
#pragma unroll
            for (int u = 0; u < PIPE; ++u) {
                const float4 qq = qt[w][idxs[u]];
                const uint32_t c = cs[u];
                const float4* lp = a.list + (uint32_t)__float_as_int(qq.w) + sub;
                const uint32_t last_line = c ? ((c - 1u) & ~7u) : 0u;
                float gd = a.sq_eps;
                int gi = -1;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {   // ascending entries: `<=` keeps the larger index on ties (IDX lists)
                    const float dx = qq.x - e0[u][e].x, dy = qq.y - e0[u][e].y, dz = qq.z - e0[u][e].z;
                    const float d = dx * dx + (dy * dy + dz * dz);
                    take_if_better<IDX>(d, __float_as_int(e0[u][e].w), gd, gi);
                }
                if (FIRST > 1 && (!NOSENT || wide[u])) {
#pragma unroll
                    for (int e = EPL; e < E0; ++e) {
                        const float dx = qq.x - e0[u][e].x, dy = qq.y - e0[u][e].y, dz = qq.z - e0[u][e].z;
                        const float d = dx * dx + (dy * dy + dz * dz);
                        take_if_better<IDX>(d, __float_as_int(e0[u][e].w), gd, gi);
                    }
                }
                if (EARLY) {
                    const float qcg = qcd[w][idxs[u]];
                    const uint32_t chunk0 = (uint32_t)__float_as_int(qq.w) >> 3;
                    uint32_t cc = c;   // entries still to be looked at (0 once the rest of the list is ruled out)
                    for (uint32_t k = 8; __any(k < cc); k += 8 * UNR) {
                        // |q - p| >= |p - centre| - |q - centre| > sqrt(best of the group) for every later p: stop
                        if (k < cc && a.chunk_r[chunk0 + (k >> 3)] - qcg > sqrtf(group_min_nonneg<GL>(gd)) + a.bound_margin) cc = 0;
                        float4 e[UNR][EPL];
#pragma unroll
                        for (int v = 0; v < UNR; ++v) {   // a line's entries under ONE condition: its loads leave together
                            if (k + 8 * v < cc) {
#pragma unroll
                                for (int x = 0; x < EPL; ++x) e[v][x] = lp[k + 8 * v + gl * x];
                            } else {
#pragma unroll
                                for (int x = 0; x < EPL; ++x) e[v][x] = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
                            }
                        }
#pragma unroll
                        for (int v = 0; v < UNR; ++v)
#pragma unroll
                            for (int x = 0; x < EPL; ++x) {
                                const float dx = qq.x - e[v][x].x, dy = qq.y - e[v][x].y, dz = qq.z - e[v][x].z;
                                const float d = dx * dx + (dy * dy + dz * dz);
                                take_if_better<IDX>(d, __float_as_int(e[v][x].w), gd, gi);
                            }
                    }
                } else
                for (uint32_t k = 8 * FIRST; __any(k < c); k += 8 * UNR) {
                    float4 e[UNR][EPL];
#pragma unroll
                    for (int v = 0; v < UNR; ++v) {
                        if (NOSENT) {   // a list shorter than this trip reads its last line again (the same entries: harmless)
                            const uint32_t kk = min(k + 8u * v, last_line);
#pragma unroll
                            for (int x = 0; x < EPL; ++x) e[v][x] = lp[kk + gl * x];
                        } else if (k + 8 * v < c) {
#pragma unroll
                            for (int x = 0; x < EPL; ++x) e[v][x] = lp[k + 8 * v + gl * x];
                        } else {
#pragma unroll
                            for (int x = 0; x < EPL; ++x) e[v][x] = make_float4(1e30f, 1e30f, 1e30f, __int_as_float(-1));
                        }
                    }
#pragma unroll
                    for (int v = 0; v < UNR; ++v)
#pragma unroll
                        for (int x = 0; x < EPL; ++x) {
                            const float dx = qq.x - e[v][x].x, dy = qq.y - e[v][x].y, dz = qq.z - e[v][x].z;
                            const float d = dx * dx + (dy * dy + dz * dz);
                            take_if_better<IDX>(d, __float_as_int(e[v][x].w), gd, gi);
                        }
                }
                const float dm = two ? group_min_nonneg<2>(gd) : group_min_nonneg<GL>(gd);
                int im = (gd == dm) ? gi : -1;
                if (GL >= 2) im = max(im, dpp_i32<DPP_QUAD_XOR1>(im));
                if (GL >= 4 && !two) im = max(im, dpp_i32<DPP_QUAD_XOR2>(im));
                if (GL == 8) im = max(im, dpp_i32<DPP_HALF_MIRROR>(im));
                if (c && sub == 0) ri[w][idxs[u]] = im;
            }
            s0 += NG * PIPE;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < nq) {
            const int idx = (head + lane) & 127;
            const int best = ri[w][idx];
            const uint32_t slot_i = qs[w][idx];
            bool counted = false;
            if (best >= 0 && STOCS_ABLATE(a, 8)) { lcp_add(acc, 0.5f); }   // 8: no normal test
            else if (best >= 0) {
                const float4 nm = STOCS_ABLATE(a, 16) ? make_float4(0.f, 0.f, 1.f, 0.f) : a.mnrm[slot_i];   // 16: no model-normal gather
                const float nx = t0 * nm.x + (t4 * nm.y + t8 * nm.z);
                const float ny = t1 * nm.x + (t5 * nm.y + t9 * nm.z);
                const float nz = t2 * nm.x + (t6 * nm.y + t10 * nm.z);
                const float4 sn = STOCS_ABLATE(a, 32) ? make_float4(nx, ny, nz, 0.5f) : snrmw[best];         // 32: no scene-normal gather
                const float d = sn.x * nx + (sn.y * ny + sn.z * nz);
                counted = (d >= a.dot_lo) && (d <= 1.0f);
                if (counted) lcp_add(acc, sn.w);
            }
            if (DETAIL) {
                const int orig = a.mperm[slot_i];
                hit_out[(size_t)cand * a.M + orig] = best;
                cnt_out[(size_t)cand * a.M + orig] = counted ? 1 : 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // one 64-point step of this wavefront's candidate: model point p of slot i
    auto step = [&](const int i, const float4 p) {
        float qx = 0.f, qy = 0.f, qz = 0.f, qcentre = 0.f, nearest = 0.f;
        uint32_t off = 0, cnt = 0;
        // slots beyond the model hold NaN positions (ctx.hip): they run through the look-up like any other query and match nothing;
        // only the per-point outputs of the detail form need the bound
        if (!DETAIL || i < a.M) {
            qx = ((t0 * p.x + t4 * p.y) + t8 * p.z) + t12;
            qy = ((t1 * p.x + t5 * p.y) + t9 * p.z) + t13;
            qz = ((t2 * p.x + t6 * p.y) + t10 * p.z) + t14;
            // one floor per axis at quarter-cell resolution gives both the cell (>> 2) and the sub-cell
            // (& 3); unsigned compares do the bounds test (a NaN query floors to 0, scans cell 0 and
            // matches nothing because every NaN distance fails d <= best)
            const int c4x = __float2int_rd((qx - a.ox) * a.inv_h4), c4y = __float2int_rd((qy - a.oy) * a.inv_h4),
                      c4z = __float2int_rd((qz - a.oz) * a.inv_h4);
            const int cx = c4x >> 2, cy = c4y >> 2, cz = c4z >> 2;
            if ((unsigned)cx < (unsigned)a.nx && (unsigned)cy < (unsigned)a.ny && (unsigned)cz < (unsigned)a.nz) {
                const int sb = ((c4z & 3) << 4) | ((c4y & 3) << 2) | (c4x & 3);
                if (FLAT) {   // one look-up: an empty cell is an all-zero word (count 0, mask 0)
                    uint4 cw = make_uint4(0u, 0u, 0u, 0u);
                    if (!STOCS_ABLATE(a, 1))   // 1: no cell-word look-up at all
                        cw = a.flat[(uint32_t)((cz * a.ny + cy) * a.nx + cx)];
                    const uint32_t mw = sb < 32 ? cw.z : cw.w;
                    off = cw.x; cnt = ((mw >> (sb & 31)) & 1u) ? cw.y : 0u;
                    if (STOCS_ABLATE(a, 2)) cnt = cw.x == 0xFFFFFFF1u ? 1u : 0u;   // 2: look-up done, nobody survives
                } else {
                    const int brick = a.top[((cz >> 3) * a.nby + (cy >> 3)) * a.nbx + (cx >> 3)];
                    if (brick >= 0) {
                        const uint4 cw = a.cells[(uint32_t)brick * 512u + (uint32_t)(((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7))];
                        if ((EARLY || NEAR) && a.has_nearest) { off = cw.x; cnt = cw.y; nearest = __uint_as_float(cw.z); }   // no mask on these grids: z = distance bound
                        else { const uint32_t mw = a.has_nearest ? 0xFFFFFFFFu : (sb < 32 ? cw.z : cw.w); off = cw.x; cnt = ((mw >> (sb & 31)) & 1u) ? cw.y : 0u; }
                    }
                }
            }
            if ((EARLY || (NEAR && a.has_nearest)) && cnt) {
                const float ex = qx - (a.ox + ((float)cx + 0.5f) * a.h), ey = qy - (a.oy + ((float)cy + 0.5f) * a.h), ez = qz - (a.oz + ((float)cz + 0.5f) * a.h);
                qcentre = sqrtf(ex * ex + (ey * ey + ez * ez));
                // every listed point is at least `nearest` from the cell centre, hence at least nearest - |q - centre| from the query:
                // beyond epsilon the list is not worth a look
                if (a.has_nearest && nearest - qcentre > a.eps + a.bound_margin) cnt = 0;
            }
            if (DETAIL && cnt == 0) {
                const int orig = a.mperm[i];
                hit_out[(size_t)cand * a.M + orig] = -1;
                cnt_out[(size_t)cand * a.M + orig] = 0;
            }
        }
        const bool hit = cnt != 0;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            if (hit) {
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                const int sl = (tail + rank) & 127;
                qt[w][sl] = make_float4(qx, qy, qz, __int_as_float((int)off));
                qn[w][sl] = cnt;
                qs[w][sl] = (uint32_t)i;
                if (EARLY) qcd[w][sl] = qcentre;
            }
            tail += __popcll(mask);
            __builtin_amdgcn_wave_barrier();
            if (tail - head >= 64) {
                process(64);
                head += 64;
            }
        }
    };
    if (TILE) {
        __shared__ float4 tile[2][64 * WPB];
        const int tid = threadIdx.x;
        const float4 pad = make_float4(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000), __int_as_float(0x7fc00000), 0.f);   // matches nothing
        tile[0][tid] = tid < a.M ? a.mpos[tid] : pad;
        __syncthreads();
        int buf = 0;
        for (int tbase = 0; tbase < a.M; tbase += 64 * WPB) {
            const int nxt = tbase + 64 * WPB + tid;                  // the next tile is requested before this one is worked on
            const float4 gn = nxt < a.M ? a.mpos[nxt] : pad;
            if (cand >= 0)
                for (int s = 0; s < WPB && tbase + 64 * s < a.M; ++s) step(tbase + 64 * s + lane, tile[buf][64 * s + lane]);
            tile[buf ^ 1][tid] = gn;
            __syncthreads();
            buf ^= 1;
        }
        if (cand < 0) return;
    } else {
    // The wavefront's steps, 64 at a time: every lane tests the patch of one step (a.patch), the ballot is the list of the
    // steps that have to be walked; the model point of the NEXT such step is requested one step ahead (takes one of the
    // three dependent loads of the look-up chain off the critical path).
    const int nsteps = (a.M + 63) >> 6;
    const int wfirst = SPLIT ? w : 0, nw = SPLIT ? WPB : 1;
    const int nk = (nsteps - wfirst + nw - 1) / nw;
    const float snorm = a.patch ? lcp_linear_norm_bound(t0, t1, t2, t4, t5, t6, t8, t9, t10) : 0.0f;
    for (int k0 = 0; k0 < nk; k0 += 64) {
        const int kk = k0 + lane;
        bool live = kk < nk;
        if (a.patch && live && !STOCS_ABLATE(a, 64))
            live = !lcp_patch_dead(a, a.patch[wfirst + kk * nw], snorm, t0, t1, t2, t4, t5, t6, t8, t9, t10, t12, t13, t14);
        unsigned long long todo = __ballot(live);
        if (DETAIL && a.patch) {   // the skipped steps' points: no neighbour, not counted
            unsigned long long dead = __ballot(kk < nk && !live);
            while (dead) {
                const int j = __builtin_ctzll(dead);
                dead &= dead - 1ull;
                const int i = ((wfirst + (k0 + j) * nw) << 6) + lane;
                if (i < a.M) {
                    const int orig = a.mperm[i];
                    hit_out[(size_t)cand * a.M + orig] = -1;
                    cnt_out[(size_t)cand * a.M + orig] = 0;
                }
            }
        }
        if (!todo) continue;
        // (the sorted positions are padded to whole steps and one step beyond: no bounds checks on these loads)
        int j = __builtin_ctzll(todo);
        float4 p_next = a.mpos[((wfirst + (k0 + j) * nw) << 6) + lane];
        for (;;) {
            const int i = ((wfirst + (k0 + j) * nw) << 6) + lane;
            const float4 p = p_next;
            todo &= todo - 1ull;
            j = todo ? __builtin_ctzll(todo) : j;   // behind the last step its own point is requested once more: no branch around the load
            p_next = a.mpos[((wfirst + (k0 + j) * nw) << 6) + lane];
            step(i, p);
            if (!todo) break;
        }
    }
    }
    if (tail - head > 0) process(tail - head);
    acc = lcp_wave_sum(acc);
    if (SPLIT) {
        if (lane == 0) part[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long total = 0;
#pragma unroll
            for (int k = 0; k < WPB; ++k) total += part[k];
            lcp_store(a, out, cand, total);
        }
    } else if (lane == 0) lcp_store(a, out, cand, acc);
}

// compute_best_transform (stocs.cpp:982-1004) on the device: max of the packed (score, ~id) keys --
// larger score wins, lower id wins ties, non-positive scores never win.  Integer max: order independent.
__global__ __launch_bounds__(256) void best_kernel(const float* __restrict__ lcp, int n, uint32_t id_offset, unsigned long long* __restrict__ best) {
    unsigned long long k = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float s = lcp[i];
        if (s > 0.0f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (id_offset + (uint32_t)i));
            k = key > k ? key : k;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0 && k) atomicMax(best, k);
}

// the same for one batch of moderate size in ONE workgroup: no zero fill in front, no atomics, the key is simply written
__global__ __launch_bounds__(1024) void best_single_kernel(const float* __restrict__ lcp, int n, uint32_t id_offset, unsigned long long* __restrict__ best) {
    __shared__ unsigned long long sh[16];
    unsigned long long k = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float s = lcp[i];
        if (s > 0.0f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (id_offset + (uint32_t)i));
            k = key > k ? key : k;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) k = sh[w] > k ? sh[w] : k;
        *best = k;
    }
}

// processing order of a batch: Morton code of the candidate's translation (= where the model centroid lands) in the
// scene grid's bounding box, 8 bits per axis (2 mm at a 0.5 m scene: candidate batches are concentrated around a few
// hypotheses, coarse buckets of 1.5 cm lost a third of the gain).  Candidates that put the model in the same place read the same
// bricks, cell words and lists; running them back to back raises the L1 / L2 hit rates (Cm: 1.67 -> 1.44 ms).
__device__ __forceinline__ uint32_t spread10(uint32_t x) {
    x &= 0x3ff; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249;
    return x;
}
__global__ __launch_bounds__(256) void order_keys_kernel(const float* __restrict__ T16, int n, float ox, float oy, float oz, float sx, float sy, float sz,
                                                         uint32_t* __restrict__ keys, int32_t* __restrict__ vals, unsigned long long* __restrict__ best) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && best) *best = 0ull;   // the arg-max word of the launch that follows (saves a fill kernel of its own)
    if (i >= n) return;
    const float* T = T16 + (size_t)i * 16;
    const float qx = fminf(fmaxf((T[12] - ox) * sx, 0.0f), 255.0f), qy = fminf(fmaxf((T[13] - oy) * sy, 0.0f), 255.0f),
                qz = fminf(fmaxf((T[14] - oz) * sz, 0.0f), 255.0f);   // NaN -> 0 (fmaxf ignores it)
    keys[i] = (spread10((uint32_t)qz) << 2) | (spread10((uint32_t)qy) << 1) | spread10((uint32_t)qx);
    vals[i] = i;
}

// Kernel selection.  The product library ships the kernels the automatic choice uses (24: scan fed from an LDS queue over
// index-ordered lists, sparse scenes; 39: the same over centre-sorted lists with early exit, dense scenes) plus two independent
// cross-checks: the per-step cooperative scan of round 2 over the centre-sorted lists (31) and the plain lane-per-query kernel
// (0); every one of them returns the reference's scores.  Everything that lost an A/B run (profiles/r01_lcp_analysis.md,
// profiles/r03_lcp_patch_and_group_ab.json: the per-step scan over index-ordered lists 15, eight lanes per query, ...) exists
// only in a tools build (make tools -> libstocs_hip_tools.so, -DSTOCS_TOOLS_BUILD), which also honours STOCS_LCP_VARIANT.
static bool lcp_variant_selectable(int v) {
    if (v == 99 || v == 0 || v == 24 || v == 31 || v == 39) return true;
#ifdef STOCS_TOOLS_BUILD
    static const int extra[] = {15, 1, 9, 16, 17, 20, 25, 26, 27, 28, 30, 32, 33, 34, 35, 40, 41, 42, 43, 44, 45, 46};
    for (size_t i = 0; i < sizeof(extra) / sizeof(extra[0]); ++i) if (extra[i] == v) return true;
#endif
    return false;
}
static int lcp_variant() {
#ifdef STOCS_TOOLS_BUILD
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("STOCS_LCP_VARIANT");
        v = (e && lcp_variant_selectable(atoi(e))) ? atoi(e) : 99;
    }
    return v;
#else
    return 99;
#endif
}

int launch_lcp(stocs_ctx* c, const float* d_T16, int n, float* d_lcp, int32_t* d_hit, uint8_t* d_counted, unsigned long long* d_best8, uint32_t id_offset) {
    if (n <= 0) { if (d_best8) STOCS_HIP_CHECK(hipMemsetAsync(d_best8, 0, 8, c->stream)); return STOCS_OK; }
    bool best_zeroed = false;
    LcpArgs a;
    a.best = d_best8; a.id_offset = id_offset;
    a.mpos = c->d_mpos_s; a.mnrm = c->d_mnrm_s; a.mperm = c->d_mperm; a.M = c->nM;
    a.top = c->grid.d_top; a.cells = c->grid.d_cells; a.flat = c->grid.d_flat; a.list = c->grid.d_list; a.snrmw = c->snrmw_override ? c->snrmw_override : c->d_snrmw;
    a.cand_trial = c->lcp_cand_trial; a.snrmw_stride = c->lcp_cand_trial ? c->snrmw_stride : 0;
    a.ox = c->grid.ox; a.oy = c->grid.oy; a.oz = c->grid.oz; a.inv_h = c->grid.inv_h; a.inv_h4 = c->grid.inv_h * 4.0f; a.h = c->grid.h; a.chunk_r = c->grid.d_chunk_r;
    a.nx = c->grid.nx; a.ny = c->grid.ny; a.nz = c->grid.nz; a.nbx = c->grid.nbx; a.nby = c->grid.nby;
    a.sq_eps = c->prm.distance_threshold * c->prm.distance_threshold;  // sq_eps = epsilon*epsilon, stocs.cpp:1014
    a.dot_lo = c->thr.lcp_dot_lo;
    a.eps = c->prm.distance_threshold;
    a.has_nearest = c->grid.has_nearest ? 1 : 0;
    {   // 4e-6 at metre scale (epsilon 5 mm, coordinates below half a metre); clouds in millimetres get a thousand times that
        double mag = 0.0;
        for (int k = 0; k < 3; ++k) mag = std::max(mag, std::max(fabs(c->grid.bb_mn[k]), fabs(c->grid.bb_mx[k])));
        a.bound_margin = (float)(4.0e-4 * (double)c->prm.distance_threshold + 4.0e-6 * mag);
    }
    // patch test: its distance field costs one pass over the scene points (0.24 ms at Cm) and takes ~6 % off a launch, so it is filled
    // once the scene has seen 1e9 point queries (three steps of the metric batch; ~25 trials of one frame) -- a caller that scores one
    // trial per frame never pays for it.  The scores do not depend on it
    a.patch = NULL; a.dist = NULL;
    a.gox = a.goy = a.goz = 0.f; a.g = a.inv_g = a.cap = 0.f; a.gnx = a.gny = a.gnz = 0;
    c->scene_scored++;
    c->scene_work += (double)n * (double)c->nM;
    if (c->lcp_cull && c->grid.d_dist && c->d_mpatch && (c->grid.dist_ready || c->lcp_cull >= 2 || c->scene_work >= c->lcp_cull_after)) {
        int rc = fill_cull_field(c);
        if (rc) return rc;
        if (c->cull_pending) { STOCS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_cull, 0)); c->cull_pending = false; }   // filled at stocs_ctx_set_scene, on the auxiliary stream
        a.patch = c->d_mpatch; a.dist = c->grid.d_dist;
        a.gox = c->grid.cg_ox; a.goy = c->grid.cg_oy; a.goz = c->grid.cg_oz; a.g = c->grid.cg_g; a.inv_g = c->grid.cg_inv_g; a.cap = c->grid.cg_cap;
        a.gnx = c->grid.cg_nx; a.gny = c->grid.cg_ny; a.gnz = c->grid.cg_nz;
    }
    const int blocks = (n + 3) / 4;
    a.order = NULL; a.xcd_blocks = 0;
#ifdef STOCS_TOOLS_BUILD
    a.ablate = getenv("STOCS_LCP_ABLATE") ? atoi(getenv("STOCS_LCP_ABLATE")) : 0;
#endif
    // big batches against scenes whose lists do not stay in the caches: spatially ordered processing (the ordering costs ~50 us; scores
    // do not depend on it).  Until the first half of round 3 it paid at Cm too (1.67 -> 1.44 ms in round 1); with the four-lane verify
    // trips it is a wash there (9 MB of lists: 1.115 ms either way at 65 536 candidates, +3 % at 32 768, +8 % for a 1 000-point model)
    // and still worth 7-12 % from 20 MB of lists on (50 000-point scene 2.55 -> 2.38 ms, C5 6.53 -> 5.73): the threshold is 12 MB;
    // lcp_order >= 2 orders whatever the size
    const bool lists_spill = (double)c->grid.n_entries * 16.0 >= 12.0e6;
    if (!d_hit && n >= 1024 && (double)n * (double)c->nM >= 1.5e8 && c->lcp_order && (lists_spill || c->lcp_order >= 2)) {
        const size_t kb = (((size_t)n * 4 + 255) / 256) * 256;
        size_t tb = 0;
        STOCS_HIP_CHECK(sort_pairs(NULL, tb, (const uint32_t*)NULL, (uint32_t*)NULL, (const uint32_t*)NULL, (uint32_t*)NULL, (size_t)n, 0, 24, c->stream));
        const size_t need = 4 * kb + tb;
        if (c->order_bytes < need) {
            if (c->d_order) { STOCS_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_order); c->d_order = NULL; c->order_bytes = 0; }
            STOCS_HIP_CHECK(dev_malloc(&c->d_order, need + need / 4));
            c->order_bytes = need + need / 4;
        }
        char* p = (char*)c->d_order;
        uint32_t* keys = (uint32_t*)p; uint32_t* keys_s = (uint32_t*)(p + kb);
        int32_t* vals = (int32_t*)(p + 2 * kb); int32_t* order = (int32_t*)(p + 3 * kb);
        void* tmp = p + 4 * kb;
        const float sx = 256.0f / ((float)c->grid.nx * c->grid.h), sy = 256.0f / ((float)c->grid.ny * c->grid.h), sz = 256.0f / ((float)c->grid.nz * c->grid.h);
        hipLaunchKernelGGL(order_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_T16, n, c->grid.ox, c->grid.oy, c->grid.oz, sx, sy, sz,
                           keys, vals, d_best8);
        best_zeroed = true;
        STOCS_HIP_CHECK(sort_pairs(tmp, tb, keys, keys_s, (const uint32_t*)vals, (uint32_t*)order, (size_t)n, 0, 24, c->stream));
        a.order = order;
        // XCD placement of the ordered list (lcp_candidate): 0 = workgroup i takes slot i, and the hardware deals consecutive workgroups to
        // the eight XCDs in turn; k > 2 = an XCD takes runs of k consecutive slots, so that what is resident on it at one time (~250
        // workgroups) comes from one or two stretches of the ordered list and shares list lines in ITS L2.  Pays only when the lists are
        // far beyond the Infinity Cache: C5 (1.6 GB) 5.73 -> 5.38 ms with runs of 128 (96: 5.60, 160: 5.95, 256: 5.49, 512 and more: 6.3
        // -- whole XCDs finish late), 65 536 candidates 20.6 -> 19.4; a 100 000-point scene (158 MB) 2.74 -> 2.98, 50 000 points 2.38 -> 2.50
        const bool lists_huge = (double)c->grid.n_entries * 16.0 >= 512.0e6;
        a.xcd_blocks = c->lcp_order >= 2 ? (c->lcp_order == 2 ? 1 : c->lcp_order) : (lists_huge ? 128 : 0);
    }
    if (d_best8 && d_best8 == c->d_best && c->best_is_zero) best_zeroed = true;   // stocs_make_transforms left the word zeroed for this launch
    if (d_best8 == c->d_best) c->best_is_zero = false;
    if (d_best8 && !best_zeroed) STOCS_HIP_CHECK(hipMemsetAsync(d_best8, 0, 8, c->stream));
    int variant = c->lcp_variant >= 0 ? c->lcp_variant : lcp_variant();
    // dense grids keep their lists sorted by distance from the cell centre (not by index): only kernels
    // instantiated with the order-independent tie rule may scan them
    const bool dense = c->grid.d_chunk_r != NULL;
    if (variant == 99)   // automatic; must not depend on the batch size (a candidate's score is batch-invariant)
        // (until the first half of round 3 sparse grids with more than 10 entries per list went to the per-step scan, variant 15; since the
        //  queue kernel verifies with four lanes per query it wins there too: 50 000 / 12 500 points 3.54 -> 2.35 ms, tools/sweep_variants.py)
        variant = dense ? 39 : 24;
    const bool dense_only = (variant >= 30 && variant <= 35) || variant == 39 || (variant >= 41 && variant <= 43);   // kernels with the order-independent tie rule / early exit
    if (dense && !(variant == 0 || variant == 16 || dense_only)) variant = 39;
    if (!dense && dense_only) variant = 24;
#define STOCS_LCP_LAUNCH(...) hipLaunchKernelGGL((__VA_ARGS__), dim3(blocks), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted)
    if (d_hit) {   // per-point detail (parity tests)
        if (dense && variant == 39) STOCS_LCP_LAUNCH(lcp_coopq_kernel<true, 1, true, 4, false, 4, 0, false, false, true>);
        else if (dense) STOCS_LCP_LAUNCH(lcp_coop_kernel<true, 2, true, true, false>);
        else if (variant == 0) STOCS_LCP_LAUNCH(lcp_kernel<true, true>);
#ifdef STOCS_TOOLS_BUILD
        else if (!(variant >= 20 && variant <= 28)) STOCS_LCP_LAUNCH(lcp_coop_kernel<true, 1, true, false, true>);
#endif
        else if (a.has_nearest) STOCS_LCP_LAUNCH(lcp_coopq_kernel<true, 1, true, 4, true, 4, 0, false, false, false, 8, 1, false, true>);   // pruned lists on a grid finer than epsilon
        else STOCS_LCP_LAUNCH(lcp_coopq_kernel<true, 1, true, 4, true>);
    } else if (dense) {
        switch (variant) {
            case 0: STOCS_LCP_LAUNCH(lcp_kernel<false, false>); break;
#ifdef STOCS_TOOLS_BUILD
            case 16: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 8, true, false, false>); break;
            case 30: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 1, true, true, false>); break;
            case 32: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 4, true, true, false>); break;
            case 33: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 2, true, true, false>); break;   // 31 with four waves per workgroup
            case 34: hipLaunchKernelGGL((lcp_coop_kernel<false, 1, true, true, false, 4, true>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); break;   // 31 with one line per trip
            case 35: hipLaunchKernelGGL((lcp_coop_kernel<false, 4, true, true, false, 4, true>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); break;   // 31 with four lines per trip
#endif
            case 39:   // queue-fed scan with early exit (16-byte lists); lanes per query as on sparse scenes (C5: 6.33 -> 5.81 ms)
#ifdef STOCS_TOOLS_BUILD
                if (c->lcp_group == 8) hipLaunchKernelGGL((lcp_coopq_kernel<false, 1, true, 4, false, 4, 0, true, false, true, 8>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted);
                else if (c->lcp_group == 42) hipLaunchKernelGGL((lcp_coopq_kernel<false, 1, true, 2, false, 4, 0, true, false, true, 4>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted);
                else if (c->lcp_group == 2) hipLaunchKernelGGL((lcp_coopq_kernel<false, 1, true, 1, false, 4, 0, true, false, true, 2>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted);
                else
#endif
                hipLaunchKernelGGL((lcp_coopq_kernel<false, 1, true, 1, false, 4, 0, true, false, true, 4>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted);
                break;
#ifdef STOCS_TOOLS_BUILD
            case 41: hipLaunchKernelGGL((lcp_coopq_kernel<false, 2, true, 4, false, 4, 0, true, false, true>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); break;   // 39, two lines per trip after the first
            case 42: hipLaunchKernelGGL((lcp_coopq_kernel<false, 1, true, 8, false, 4, 0, true, false, true>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); break;   // 39, eight first lines in flight
            case 43: hipLaunchKernelGGL((lcp_coopq_kernel<false, 1, false, 4, false, 4, 0, true, false, true>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); break;  // 39 without the ordering by list length
#endif
            default:   // 31
                // four wavefronts per candidate: C5 (16 384 candidates x 50 000 points) 11.6 -> 8.3 ms; eight: 8.1 ms, but 20 % slower at Cm
                if (c->lcp_split && a.M >= 512) hipLaunchKernelGGL((lcp_coop_kernel<false, 2, true, true, false, 4, true>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted);
                else hipLaunchKernelGGL((lcp_coop_kernel<false, 2, true, true, false, 1>), dim3(n), dim3(64), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted);
                break;
        }
    } else {
        switch (variant) {
            case 0: STOCS_LCP_LAUNCH(lcp_kernel<false, true>); break;
#ifdef STOCS_TOOLS_BUILD
            case 15: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 4, true, false, true>); break;
            case 1: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 1, true, false, true>); break;
            case 9: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 2, true, false, true>); break;
            case 16: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 8, true, false, true>); break;
            case 17: STOCS_LCP_LAUNCH(lcp_coop_kernel<false, 2, false, false, true>); break;   // no sub-cell mask
            case 20: STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, false, 4, true>); break;
            case 26: STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 1, true>); break;
            case 27: STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 2, true>); break;
            case 28: STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 8, true>); break;
            case 25: STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 2, true, 4, true>); break;
            // waves per workgroup (the kernel has no workgroup-wide barrier): 16 / 8 / 4 / 2 / 1 -> 1.83 / 1.62 / 1.51 / 1.51 / 1.475 ms at Cm
            case 40: hipLaunchKernelGGL((lcp_coopq_kernel<false, 1, true, 4, true, 8>), dim3((n + 7) / 8), dim3(512), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); break;
            case 44: STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 4, true, 4>); break;
            // model tile shared through LDS by the four candidates of a workgroup (45), next to the same workgroup shape without it (46)
            case 45: if (a.flat) STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 4, true, 4, 1, false, true>); else STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 4, true, 4, 0, false, true>); break;
            case 46: if (a.flat) STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 4, true, 4, 1>); else STOCS_LCP_LAUNCH(lcp_coopq_kernel<false, 1, true, 4, true, 4>); break;
#endif
            default: {  // 24
                const int flat = (c->lcp_flat && a.flat) ? 1 : 0;
                // four wavefronts per candidate: a trial's ~8 000 candidates finish 40 % sooner (one round of long wavefronts
                // becomes four rounds of short ones), 32 768 candidates 9 % sooner, 65 536 the same (tools/lcp_flat_ab.py)
                const bool split = c->lcp_split && a.M >= 512;
#define STOCS_LCP_Q(SORTV, PIPEV, GLV, UNRV, ...) do { \
                    if (split) { if (flat) hipLaunchKernelGGL((lcp_coopq_kernel<false, UNRV, SORTV, PIPEV, true, 4, 1, true, false, false, GLV, ##__VA_ARGS__>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); \
                                 else hipLaunchKernelGGL((lcp_coopq_kernel<false, UNRV, SORTV, PIPEV, true, 4, 0, true, false, false, GLV, ##__VA_ARGS__>), dim3(n), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); } \
                    else { if (flat) hipLaunchKernelGGL((lcp_coopq_kernel<false, UNRV, SORTV, PIPEV, true, 1, 1, false, false, false, GLV, ##__VA_ARGS__>), dim3(n), dim3(64), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); \
                           else hipLaunchKernelGGL((lcp_coopq_kernel<false, UNRV, SORTV, PIPEV, true, 1, 0, false, false, false, GLV, ##__VA_ARGS__>), dim3(n), dim3(64), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted); } } while (0)
                // lanes per queued query in the verify trips (stocs_set_option "lcp_group"; profiles/r03_lcp_group_ab.json): 4 lanes with
                // two list entries each and one trip (16 queries) in flight -- Cm 1.34 -> 1.18 ms; 8 lanes with one entry each and four
                // trips in flight is the form of rounds 1-3 (tools build)
                switch (c->lcp_group) {
#ifdef STOCS_TOOLS_BUILD
                    case 8: STOCS_LCP_Q(true, 4, 8, 1); break;       // eight lanes with one entry each, four trips in flight: the form of rounds 1-3a
                    case 42: STOCS_LCP_Q(true, 2, 4, 1); break;    // 4 lanes, two trips in flight (1.185 ms)
                    case 41: STOCS_LCP_Q(true, 1, 4, 1, 1); break; // 4 with one line in the first trip (+2 % at 65 536 candidates, +7 % at 8 192)
                    case 49: STOCS_LCP_Q(true, 1, 4, 1, 2, false); break; // 4 with predicated list loads and sentinel entries (+1.2 %; +2.4 % at 8 192)
                    case 46: STOCS_LCP_Q(true, 1, 4, 1, 3); break; // ... three lines (+3 %)
                    case 47: STOCS_LCP_Q(true, 1, 4, 2, 2); break; // 4 with two lines per later trip as well (+3 %)
                    case 48: STOCS_LCP_Q(true, 1, 4, 2, 1); break; // one line first, then two per trip (+5 %)
                    case 85: STOCS_LCP_Q(true, 2, 8, 1, 2); break; // 8 lanes, two trips, two lines
                    case 43: STOCS_LCP_Q(true, 3, 4, 1); break;    // ... three (1.35 ms: spills)
                    case 44: STOCS_LCP_Q(true, 4, 4, 1); break;    // ... four (2.56 ms: spills)
                    case 40: STOCS_LCP_Q(false, 2, 4, 1); break;   // 42 without the ordering by list length (1.24 ms)
                    case 2: STOCS_LCP_Q(true, 1, 2, 1); break;     // 2 lanes, four entries each (1.45 ms)
                    case 22: STOCS_LCP_Q(true, 2, 2, 1); break;    // ... two trips in flight (2.11 ms)
                    case 1: STOCS_LCP_Q(true, 1, 1, 1); break;     // a lane per query, a whole line per lane (3.16 ms)
#endif
                    default:
                        // four lanes per query, one trip in flight, a list's first two lines together, no sentinels.  has_nearest: pruned lists on a grid
                        // finer than epsilon, the distance bound in place of the sub-cell mask.  (Round 5 measured and did not keep TINY -- lists of at
                        // most four entries verified by two lanes, 32 queries per trip: Cm 0.939 -> 1.001 ms, C5 3.30 -> 3.50 ms, scores bitwise equal:
                        // the runtime lane mapping and the fourth ordering class cost more instructions than the saved addresses are worth.  The
                        // form lives in the tools build, STOCS_LCP_TINY=1.)
#ifdef STOCS_TOOLS_BUILD
                        if (getenv("STOCS_LCP_TINY")) { if (a.has_nearest) STOCS_LCP_Q(true, 1, 4, 1, 2, true, true, true); else STOCS_LCP_Q(true, 1, 4, 1, 2, true, false, true); break; }
#endif
                        if (a.has_nearest) STOCS_LCP_Q(true, 1, 4, 1, 2, true, true); else STOCS_LCP_Q(true, 1, 4, 1, 2, true);
                        break;
                }
#undef STOCS_LCP_Q
                break;
            }
        }
    }
#undef STOCS_LCP_LAUNCH
    STOCS_HIP_CHECK(hipGetLastError());
    return STOCS_OK;
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_score_transforms_device(stocs_ctx* c, const void* d_T16, int n, void* d_lcp) {
    if (!c || n < 0 || (n && (!d_T16 || !d_lcp))) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    return launch_lcp(c, (const float*)d_T16, n, (float*)d_lcp, NULL, NULL, NULL, 0);
}

int stocs_score_transforms(stocs_ctx* c, const float* T_host, int n, float* lcp_host) {
    if (!c || n < 0 || (n && (!T_host || !lcp_host))) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (n == 0) return STOCS_OK;
    const size_t tb = (size_t)n * 64, lb = (size_t)n * 4;
    int rc = ensure_scratch(c, tb + lb + 256);
    if (rc) return rc;
    float* dT = (float*)c->d_scratch;
    float* dL = (float*)((char*)c->d_scratch + ((tb + 255) / 256) * 256);
    STOCS_HIP_CHECK(hipMemcpyAsync(dT, T_host, tb, hipMemcpyHostToDevice, c->stream));
    rc = launch_lcp(c, dT, n, dL, NULL, NULL, NULL, 0);
    if (rc) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(lcp_host, dL, lb, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

int stocs_lcp_detail(stocs_ctx* c, const float* T_host, int32_t* hit, uint8_t* counted) {
    if (!c || !T_host || !hit || !counted) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    const size_t M = (size_t)c->nM;
    int rc = ensure_scratch(c, 256 + 256 + M * 4 + 256 + M);
    if (rc) return rc;
    char* base = (char*)c->d_scratch;
    float* dT = (float*)base;
    float* dL = (float*)(base + 256);
    int32_t* dH = (int32_t*)(base + 512);
    uint8_t* dC = (uint8_t*)(base + 512 + ((M * 4 + 255) / 256) * 256);
    STOCS_HIP_CHECK(hipMemcpyAsync(dT, T_host, 64, hipMemcpyHostToDevice, c->stream));
    rc = launch_lcp(c, dT, 1, dL, dH, dC, NULL, 0);
    if (rc) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(hit, dH, M * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(counted, dC, M, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

// hits / counted flags of a chunk of candidates -> two totals
__global__ __launch_bounds__(256) void hit_count_kernel(const int32_t* __restrict__ hit, const uint8_t* __restrict__ counted, size_t n, unsigned long long* __restrict__ out2) {
    unsigned long long h = 0, k = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { h += hit[i] >= 0 ? 1u : 0u; k += counted[i] ? 1u : 0u; }
    for (int off = 32; off > 0; off >>= 1) { h += __shfl_xor(h, off, 64); k += __shfl_xor(k, off, 64); }
    if ((threadIdx.x & 63) == 0) { if (h) atomicAdd(&out2[0], h); if (k) atomicAdd(&out2[1], k); }
}

int stocs_lcp_hit_count(stocs_ctx* c, const void* d_T16, int n, int64_t* hits, int64_t* counted) {
    if (!c || n < 0 || (n && !d_T16) || !hits) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    *hits = 0;
    if (counted) *counted = 0;
    if (n == 0 || c->nM == 0) return STOCS_OK;
    const size_t M = (size_t)c->nM;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)256 << 20) / (5 * M)));
    const size_t hb = (((size_t)chunk * M * 4 + 255) / 256) * 256, cb = (((size_t)chunk * M + 255) / 256) * 256, lb = (((size_t)chunk * 4 + 255) / 256) * 256;
    int rc = ensure_scratch(c, 256 + lb + hb + cb);
    if (rc) return rc;
    char* base = (char*)c->d_scratch;
    unsigned long long* d_out = (unsigned long long*)base;
    float* dL = (float*)(base + 256);
    int32_t* dH = (int32_t*)(base + 256 + lb);
    uint8_t* dC = (uint8_t*)(base + 256 + lb + hb);
    STOCS_HIP_CHECK(hipMemsetAsync(d_out, 0, 16, c->stream));
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        rc = launch_lcp(c, (const float*)d_T16 + (size_t)i0 * 16, m, dL, dH, dC, NULL, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(hit_count_kernel, dim3(1024), dim3(256), 0, c->stream, (const int32_t*)dH, (const uint8_t*)dC, (size_t)m * M, d_out);
        STOCS_HIP_CHECK(hipGetLastError());
    }
    if ((rc = ensure_pinned(c, PIN_VAR))) return rc;
    unsigned long long* pin = (unsigned long long*)((char*)c->h_pin + PIN_BEST);
    STOCS_HIP_CHECK(hipMemcpyAsync(pin, d_out, 16, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *hits = (int64_t)pin[0];
    if (counted) *counted = (int64_t)pin[1];
    return STOCS_OK;
}

int stocs_best_device(stocs_ctx* c, const void* d_lcp, int n, uint32_t id_offset, uint64_t* key) {
    if (!c || !key || n < 0 || (n && !d_lcp)) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    *key = 0;
    if (n == 0) return STOCS_OK;
    if (!c->d_best) STOCS_HIP_CHECK(dev_malloc((void**)&c->d_best, 8));
    c->best_is_zero = false;
    if (n <= (1 << 18)) {
        hipLaunchKernelGGL(best_single_kernel, dim3(1), dim3(1024), 0, c->stream, (const float*)d_lcp, n, id_offset, c->d_best);
    } else {
        STOCS_HIP_CHECK(hipMemsetAsync(c->d_best, 0, 8, c->stream));
        const int blocks = std::min((n + 255) / 256, 1024);
        hipLaunchKernelGGL(best_kernel, dim3(blocks), dim3(256), 0, c->stream, (const float*)d_lcp, n, id_offset, c->d_best);
    }
    STOCS_HIP_CHECK(hipGetLastError());
    { int rc = ensure_pinned(c, PIN_VAR); if (rc) return rc; }
    uint64_t* key_pin = (uint64_t*)((char*)c->h_pin + PIN_BEST);
    STOCS_HIP_CHECK(hipMemcpyAsync(key_pin, c->d_best, 8, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *key = *key_pin;
    return STOCS_OK;
}

int stocs_best_device_async(stocs_ctx* c, const void* d_lcp, int n, uint32_t id_offset, void* d_key8) {
    if (!c || !d_key8 || n < 0 || (n && !d_lcp)) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (n == 0) { STOCS_HIP_CHECK(hipMemsetAsync(d_key8, 0, 8, c->stream)); return STOCS_OK; }
    if (n <= (1 << 18)) {
        hipLaunchKernelGGL(best_single_kernel, dim3(1), dim3(1024), 0, c->stream, (const float*)d_lcp, n, id_offset, (unsigned long long*)d_key8);
    } else {
        STOCS_HIP_CHECK(hipMemsetAsync(d_key8, 0, 8, c->stream));
        const int blocks = std::min((n + 255) / 256, 1024);
        hipLaunchKernelGGL(best_kernel, dim3(blocks), dim3(256), 0, c->stream, (const float*)d_lcp, n, id_offset, (unsigned long long*)d_key8);
    }
    STOCS_HIP_CHECK(hipGetLastError());
    return STOCS_OK;
}

int stocs_score_best_device_async(stocs_ctx* c, const void* d_T16, int n, void* d_lcp, uint32_t id_offset, void* d_key8) {
    if (!c || !d_key8 || n < 0 || (n && (!d_T16 || !d_lcp))) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    return launch_lcp(c, (const float*)d_T16, n, (float*)d_lcp, NULL, NULL, (unsigned long long*)d_key8, id_offset);
}

int stocs_score_best_device(stocs_ctx* c, const void* d_T16, int n, void* d_lcp, uint32_t id_offset, uint64_t* key) {
    if (!c || !key) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->d_best) STOCS_HIP_CHECK(dev_malloc((void**)&c->d_best, 8));
    int rc = stocs_score_best_device_async(c, d_T16, n, d_lcp, id_offset, c->d_best);
    if (rc) return rc;
    if ((rc = ensure_pinned(c, PIN_VAR))) return rc;
    uint64_t* key_pin = (uint64_t*)((char*)c->h_pin + PIN_BEST);
    STOCS_HIP_CHECK(hipMemcpyAsync(key_pin, c->d_best, 8, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *key = *key_pin;
    return STOCS_OK;
}

int stocs_set_option(stocs_ctx* c, const char* key, int value) {
    if (!c || !key) return STOCS_ERR_INVALID;
    if (!strcmp(key, "lcp_variant")) {
        if (!lcp_variant_selectable(value)) { set_error("stocs_set_option: lcp_variant %d is not part of this build (99 automatic, 0, 24, 31, 39)", value); return STOCS_ERR_INVALID; }
        c->lcp_variant = value;
        return STOCS_OK;
    }
    // 0 off, 1 spatial order, 2 + XCD-contiguous halves of the list, k > 2 + chunks of k consecutive slots per XCD
    if (!strcmp(key, "lcp_order") && value >= 0 && value <= 4096) { c->lcp_order = value; return STOCS_OK; }
    // 0: brick look-ups only, 1: the flat cell table when the grid has one (takes effect for kernels launched afterwards;
    // the table itself is built with the scene grid)
    // 1: stocs_find_congruent_all times its kernel groups with HIP events ("device: ..." steps of stocs_last_call_timing); 0 (default): host steps only
    if (!strcmp(key, "device_clock") && (value == 0 || value == 1)) { c->device_clock = value; return STOCS_OK; }
    if (!strcmp(key, "lcp_flat") && (value == 0 || value == 1)) { c->lcp_flat = value; return STOCS_OK; }
    // 0: one wavefront per candidate, 1 (default): four wavefronts share a candidate's model points (same scores)
    if (!strcmp(key, "lcp_split") && (value == 0 || value == 1)) { c->lcp_split = value; return STOCS_OK; }
    // 0: every 64-point step is walked; 1 (default): steps whose bounding sphere is out of reach of the scene are skipped once the scene's
    // distance field pays (1e9 point queries against the scene so far); 2: from the first call (same scores in every case)
    if (!strcmp(key, "lcp_cull") && value >= 0 && value <= 2) { c->lcp_cull = value; return STOCS_OK; }
    // the threshold of lcp_cull = 1, in MILLIONS of point queries (candidates x model points) scored against the current scene:
    // default 1000 (= 1e9: the field costs ~0.25 ms at 20 000 scene points and takes ~6 % off a launch); 0 = from the first call
    if (!strcmp(key, "lcp_cull_after") && value >= 0) { c->lcp_cull_after = (double)value * 1.0e6; return STOCS_OK; }
    // lanes that verify one queued query together: 4 (two list entries per lane); the eight-lane form of rounds 1-3a lost its A/B and lives in the tools build
    if (!strcmp(key, "lcp_group")) {
        bool ok = value == 4;
#ifdef STOCS_TOOLS_BUILD
        ok = ok || value == 8 || value == 41 || value == 49 || value == 46 || value == 47 || value == 48 || value == 85 || value == 42 || value == 43 || value == 44 || value == 40 || value == 2 || value == 22 || value == 1;
#endif
        if (ok) { c->lcp_group = value; return STOCS_OK; }
    }
    set_error("stocs_set_option: unknown option or value");
    return STOCS_ERR_INVALID;
}

int stocs_get_cull_state(stocs_ctx* c, float* patches4, int32_t* perm, int* n_patches, float* geom8, float* dist, int64_t dist_cap, int64_t* n_dist) {
    if (!c || !n_patches || !n_dist) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    const int np = c->d_mpatch ? (c->nM + 63) / 64 : 0;
    *n_patches = np;
    const SceneGrid& g = c->grid;
    const int64_t nd = g.d_dist ? (int64_t)g.cg_nx * g.cg_ny * g.cg_nz : 0;
    *n_dist = nd;
    // (everything that can fail comes before the first copy into the caller's pageable buffers: no copy is left in flight on an error return)
    if (dist && nd) {
        if (dist_cap < nd) return STOCS_ERR_CAPACITY;
        int rc = fill_cull_field(c);
        if (rc) return rc;
        if (c->cull_pending) { STOCS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_cull, 0)); c->cull_pending = false; }
    }
    if (patches4 && np) STOCS_HIP_CHECK(hipMemcpyAsync(patches4, c->d_mpatch, (size_t)np * 16, hipMemcpyDeviceToHost, c->stream));
    if (perm) for (int i = 0; i < c->nM; ++i) perm[i] = c->h_mperm[i];
    if (geom8) {
        geom8[0] = g.cg_ox; geom8[1] = g.cg_oy; geom8[2] = g.cg_oz; geom8[3] = g.cg_g; geom8[4] = g.cg_cap;
        geom8[5] = (float)g.cg_nx; geom8[6] = (float)g.cg_ny; geom8[7] = (float)g.cg_nz;
    }
    hipError_t e = hipSuccess;
    if (dist && nd) e = hipMemcpyAsync(dist, g.d_dist, (size_t)nd * 4, hipMemcpyDeviceToHost, c->stream);
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));   // also behind a failed enqueue: the earlier copies have landed before the caller's buffers go away
    STOCS_HIP_CHECK(e);
    return STOCS_OK;
}

int stocs_time_score_kernel(stocs_ctx* c, const void* d_T16, int n, void* d_lcp, int reps, float* avg_ms) {
    if (!c || !avg_ms || reps <= 0 || n <= 0) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    STOCS_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; ++r) {
        int rc = launch_lcp(c, (const float*)d_T16, n, (float*)d_lcp, NULL, NULL, NULL, 0);
        if (rc) return rc;
    }
    STOCS_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    STOCS_HIP_CHECK(hipEventSynchronize(c->ev1));
    float ms = 0;
    STOCS_HIP_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = ms / (float)reps;
    return STOCS_OK;
}

}  // extern "C"
