// lcp.hip -- batched weighted-LCP verification, THE METRIC KERNEL ("candidate poses verified / s").
// Replaces stocs_estimator::compute_alignment_score_for_rigid_transform (reference
// src/stocs.cpp:1006-1041) and the kd-tree query it calls per model point
// (reference include/super4pcs/accelerators/kdtree.h:394-459).
//
// Mapping: one 64-lane wavefront per candidate transform, four candidates per 256-thread workgroup.
// The 3x4 transform is wave-uniform (scalar loads -> SGPRs).  Lane l walks the Morton-sorted model
// points l, l+64, ...: coalesced 16-byte loads, and the 64 queries of one step fall into a handful
// of neighbouring grid cells, so the brick/cell/list gathers of a wave share cache lines.
// Per query: transform the point, locate its cell (top -> brick -> cell word), scan that cell's
// candidate list (every scene point within epsilon of the cell box) for the nearest point with
// d^2 <= epsilon^2, then the 30-degree normal test as an exact threshold on the dot product, and a
// wave butterfly reduction of the class-probability weights.  No atomics: results are run-to-run
// deterministic.
//
// Roofline: HBM-read model, algorithmic bytes 68 + 52*|M| per pose.
#include "stocs_ctx.h"

namespace stocs {

struct LcpArgs {
    const float4* mpos;   // Morton-sorted centred model positions
    const float4* mnrm;
    const int32_t* mperm; // sorted slot -> original model index (detail output only)
    int M;
    const int32_t* top;
    const uint2* cells;
    const float4* list;
    const float4* snrmw;  // scene unit normal + class-probability weight
    float ox, oy, oz, inv_h;
    int nx, ny, nz, nbx, nby;
    float sq_eps, dot_lo;
};

template <bool DETAIL>
__global__ __launch_bounds__(256) void lcp_kernel(LcpArgs a, const float* __restrict__ T16, float* __restrict__ out,
                                                  int n, int32_t* __restrict__ hit_out, uint8_t* __restrict__ cnt_out) {
    const int lane = threadIdx.x & 63;
    const int cand = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (cand >= n) return;
    const float* T = T16 + (size_t)cand * 16;
    const float t0 = T[0], t1 = T[1], t2 = T[2], t4 = T[4], t5 = T[5], t6 = T[6], t8 = T[8], t9 = T[9], t10 = T[10],
                t12 = T[12], t13 = T[13], t14 = T[14];
    float acc = 0.0f;
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
                const uint2 cw = a.cells[(size_t)brick * 512 + (((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7))];
                float bd = a.sq_eps;
                for (uint32_t k = 0; k < cw.y; ++k) {
                    const float4 s = a.list[cw.x + k];
                    const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
                    const float d = dx * dx + (dy * dy + dz * dz);
                    if (d <= bd) { bd = d; best = __float_as_int(s.w); }  // inclusive radius (kdtree.h:424)
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
            const float4 sn = a.snrmw[best];
            const float d = sn.x * nx + (sn.y * ny + sn.z * nz);
            // acos(d)*180/pi < 30 as an exact threshold; d > 1 -> NaN angle -> not counted (Q7)
            counted = (d >= a.dot_lo) && (d <= 1.0f);
            if (counted) acc += sn.w;
        }
        if (DETAIL) {
            const int orig = a.mperm[i];
            hit_out[(size_t)cand * a.M + orig] = best;
            cnt_out[(size_t)cand * a.M + orig] = counted ? 1 : 0;
        }
    }
    // fixed-shape butterfly: deterministic
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[cand] = acc / (float)a.M;
}

int launch_lcp(stocs_ctx* c, const float* d_T16, int n, float* d_lcp, int32_t* d_hit, uint8_t* d_counted) {
    if (n <= 0) return STOCS_OK;
    LcpArgs a;
    a.mpos = c->d_mpos_s; a.mnrm = c->d_mnrm_s; a.mperm = c->d_mperm; a.M = c->nM;
    a.top = c->grid.d_top; a.cells = c->grid.d_cells; a.list = c->grid.d_list; a.snrmw = c->d_snrmw;
    a.ox = c->grid.ox; a.oy = c->grid.oy; a.oz = c->grid.oz; a.inv_h = c->grid.inv_h;
    a.nx = c->grid.nx; a.ny = c->grid.ny; a.nz = c->grid.nz; a.nbx = c->grid.nbx; a.nby = c->grid.nby;
    a.sq_eps = c->prm.distance_threshold * c->prm.distance_threshold;  // sq_eps = epsilon*epsilon, stocs.cpp:1014
    a.dot_lo = c->thr.lcp_dot_lo;
    const int blocks = (n + 3) / 4;
    if (d_hit)
        hipLaunchKernelGGL(lcp_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, d_hit, d_counted);
    else
        hipLaunchKernelGGL(lcp_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, a, d_T16, d_lcp, n, (int32_t*)NULL, (uint8_t*)NULL);
    STOCS_HIP_CHECK(hipGetLastError());
    return STOCS_OK;
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_score_transforms_device(stocs_ctx* c, const void* d_T16, int n, void* d_lcp) {
    if (!c || n < 0 || (n && (!d_T16 || !d_lcp))) return STOCS_ERR_INVALID;
    return launch_lcp(c, (const float*)d_T16, n, (float*)d_lcp, NULL, NULL);
}

int stocs_score_transforms(stocs_ctx* c, const float* T_host, int n, float* lcp_host) {
    if (!c || n < 0 || (n && (!T_host || !lcp_host))) return STOCS_ERR_INVALID;
    if (n == 0) return STOCS_OK;
    const size_t tb = (size_t)n * 64, lb = (size_t)n * 4;
    int rc = ensure_scratch(c, tb + lb + 256);
    if (rc) return rc;
    float* dT = (float*)c->d_scratch;
    float* dL = (float*)((char*)c->d_scratch + ((tb + 255) / 256) * 256);
    STOCS_HIP_CHECK(hipMemcpyAsync(dT, T_host, tb, hipMemcpyHostToDevice, c->stream));
    rc = launch_lcp(c, dT, n, dL, NULL, NULL);
    if (rc) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(lcp_host, dL, lb, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

int stocs_lcp_detail(stocs_ctx* c, const float* T_host, int32_t* hit, uint8_t* counted) {
    if (!c || !T_host || !hit || !counted) return STOCS_ERR_INVALID;
    const size_t M = (size_t)c->nM;
    int rc = ensure_scratch(c, 256 + 256 + M * 4 + 256 + M);
    if (rc) return rc;
    char* base = (char*)c->d_scratch;
    float* dT = (float*)base;
    float* dL = (float*)(base + 256);
    int32_t* dH = (int32_t*)(base + 512);
    uint8_t* dC = (uint8_t*)(base + 512 + ((M * 4 + 255) / 256) * 256);
    STOCS_HIP_CHECK(hipMemcpyAsync(dT, T_host, 64, hipMemcpyHostToDevice, c->stream));
    rc = launch_lcp(c, dT, 1, dL, dH, dC);
    if (rc) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(hit, dH, M * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(counted, dC, M, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

int stocs_time_score_kernel(stocs_ctx* c, const void* d_T16, int n, void* d_lcp, int reps, float* avg_ms) {
    if (!c || !avg_ms || reps <= 0 || n <= 0) return STOCS_ERR_INVALID;
    STOCS_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; ++r) {
        int rc = launch_lcp(c, (const float*)d_T16, n, (float*)d_lcp, NULL, NULL);
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
