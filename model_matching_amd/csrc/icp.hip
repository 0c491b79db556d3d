// icp.hip -- pose refinement, SURVEY.md 8(f) row 4: clustering::point_to_plane_icp (reference
// src/pose_clustering.cpp:123-140), a wrapper around pcl::IterativeClosestPointWithNormals with
// 5 iterations and a 3.5 cm correspondence distance.  It has no caller in the reference and its
// arithmetic lives in PCL (absent): PARITY WITH THE REFERENCE IS UNPINNED.  This is the textbook
// algorithm PCL implements (nearest-neighbour correspondences within the distance, linearised
// point-to-plane least squares, Rz*Ry*Rx update), pinned against oracle/ingest_oracle.py::icp.
//
// Per iteration one kernel: each thread owns a source point, the target cloud streams through LDS in
// 256-point tiles (coalesced float4 loads), the thread keeps its nearest target, then the block reduces
// the 27 unique entries of A^T A | A^T b in double and writes one partial per block (summed on the host in
// block order: deterministic).  The 6x6 solve is host code.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "stocs_ctx.h"

namespace stocs {

__global__ __launch_bounds__(256) void icp_accumulate_kernel(const float4* __restrict__ src, int nsrc, const float4* __restrict__ tpos,
                                                             const float4* __restrict__ tnrm, int ntgt, const double* __restrict__ T /*row-major 3x4*/,
                                                             double max_d2, double* __restrict__ partial /*gridDim.x * 28*/) {
    __shared__ float4 tile[256];
    __shared__ double red[256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double sx = 0, sy = 0, sz = 0;
    if (i < nsrc) {
        const float4 s = src[i];
        sx = T[0] * s.x + T[1] * s.y + T[2] * s.z + T[3];
        sy = T[4] * s.x + T[5] * s.y + T[6] * s.z + T[7];
        sz = T[8] * s.x + T[9] * s.y + T[10] * s.z + T[11];
    }
    double best = 1e300;
    int bj = -1;
    for (int base = 0; base < ntgt; base += 256) {
        if (base + (int)threadIdx.x < ntgt) tile[threadIdx.x] = tpos[base + threadIdx.x];
        __syncthreads();
        const int m = min(256, ntgt - base);
        if (i < nsrc)
            for (int k = 0; k < m; ++k) {
                const double dx = sx - tile[k].x, dy = sy - tile[k].y, dz = sz - tile[k].z;
                const double d = dx * dx + dy * dy + dz * dz;
                if (d < best) { best = d; bj = base + k; }   // first minimum wins
            }
        __syncthreads();
    }
    double v[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) v[k] = 0;
    if (i < nsrc && bj >= 0 && best <= max_d2) {
        const float4 t = tpos[bj], n = tnrm[bj];
        const double a[6] = {sy * n.z - sz * n.y, sz * n.x - sx * n.z, sx * n.y - sy * n.x, n.x, n.y, n.z};   // [s x n, n]
        const double b = (t.x - sx) * n.x + (t.y - sy) * n.y + (t.z - sz) * n.z;
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) v[k++] = a[r] * a[c];
#pragma unroll
        for (int r = 0; r < 6; ++r) v[21 + r] = a[r] * b;
        v[27] = 1.0;   // number of correspondences
    }
    for (int k = 0; k < 28; ++k) {
        red[threadIdx.x] = v[k];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * 28 + k] = red[0];
        __syncthreads();
    }
}

static bool solve6(double A[6][6], double b[6], double x[6]) {
    int piv[6];
    for (int i = 0; i < 6; ++i) piv[i] = i;
    for (int c = 0; c < 6; ++c) {
        int p = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (fabs(A[p][c]) < 1e-300) return false;
        if (p != c) { for (int k = 0; k < 6; ++k) std::swap(A[p][k], A[c][k]); std::swap(b[p], b[c]); }
        for (int r = c + 1; r < 6; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 6; ++k) A[r][k] -= f * A[c][k];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < 6; ++k) s -= A[r][k] * x[k];
        x[r] = s / A[r][r];
    }
    return true;
}

}  // namespace stocs

using namespace stocs;

extern "C" int stocs_icp_point_to_plane(const float* src_pos3, int nsrc, const float* tgt_pos3, const float* tgt_nrm3, int ntgt,
                                        int max_iterations, float max_correspondence_distance, int device, float* T16_out,
                                        int* n_correspondences) {
    if (!src_pos3 || !tgt_pos3 || !tgt_nrm3 || !T16_out || nsrc <= 0 || ntgt <= 0 || max_iterations < 0) return STOCS_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available: this library has no CPU fallback"); return STOCS_ERR_NO_DEVICE; }
    if (device >= 0) STOCS_HIP_CHECK(hipSetDevice(device));
    std::vector<float4> hs((size_t)nsrc), ht((size_t)ntgt), hn((size_t)ntgt);
    for (int i = 0; i < nsrc; ++i) hs[i] = make_float4(src_pos3[3 * i], src_pos3[3 * i + 1], src_pos3[3 * i + 2], 0.f);
    for (int i = 0; i < ntgt; ++i) {
        ht[i] = make_float4(tgt_pos3[3 * i], tgt_pos3[3 * i + 1], tgt_pos3[3 * i + 2], 0.f);
        hn[i] = make_float4(tgt_nrm3[3 * i], tgt_nrm3[3 * i + 1], tgt_nrm3[3 * i + 2], 0.f);
    }
    float4 *ds = NULL, *dt = NULL, *dn = NULL;
    double *dT = NULL, *dP = NULL;
    const int blocks = (nsrc + 255) / 256;
    STOCS_HIP_CHECK(dev_malloc((void**)&ds, sizeof(float4) * (size_t)nsrc));
    STOCS_HIP_CHECK(dev_malloc((void**)&dt, sizeof(float4) * (size_t)ntgt));
    STOCS_HIP_CHECK(dev_malloc((void**)&dn, sizeof(float4) * (size_t)ntgt));
    STOCS_HIP_CHECK(dev_malloc((void**)&dT, sizeof(double) * 12));
    STOCS_HIP_CHECK(dev_malloc((void**)&dP, sizeof(double) * 28 * (size_t)blocks));
    STOCS_HIP_CHECK(hipMemcpy(ds, hs.data(), sizeof(float4) * (size_t)nsrc, hipMemcpyHostToDevice));
    STOCS_HIP_CHECK(hipMemcpy(dt, ht.data(), sizeof(float4) * (size_t)ntgt, hipMemcpyHostToDevice));
    STOCS_HIP_CHECK(hipMemcpy(dn, hn.data(), sizeof(float4) * (size_t)ntgt, hipMemcpyHostToDevice));
    double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};   // row-major 3x4, source -> target
    std::vector<double> part((size_t)blocks * 28);
    int ncorr = 0;
    int rc = STOCS_OK;
    for (int it = 0; it < max_iterations; ++it) {
        if (hipMemcpy(dT, T, sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { rc = STOCS_ERR_HIP; break; }
        hipLaunchKernelGGL(icp_accumulate_kernel, dim3(blocks), dim3(256), 0, 0, ds, nsrc, dt, dn, ntgt, dT,
                           (double)max_correspondence_distance * (double)max_correspondence_distance, dP);
        if (hipGetLastError() != hipSuccess || hipMemcpy(part.data(), dP, sizeof(double) * part.size(), hipMemcpyDeviceToHost) != hipSuccess) { rc = STOCS_ERR_HIP; break; }
        double acc[28] = {0};
        for (int b = 0; b < blocks; ++b) for (int k = 0; k < 28; ++k) acc[k] += part[(size_t)b * 28 + k];
        ncorr = (int)acc[27];
        if (ncorr < 6) break;   // not enough correspondences: keep the current estimate
        double A[6][6], bb[6], x[6];
        int k = 0;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { A[r][c] = acc[k]; A[c][r] = acc[k]; k++; }
        for (int r = 0; r < 6; ++r) bb[r] = acc[21 + r];
        if (!solve6(A, bb, x)) break;
        // update = [Rz(gamma) Ry(beta) Rx(alpha) | t], composed on the left of the running estimate
        const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
        const double R[3][3] = {{cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa},
                                {sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa},
                                {-sb, cb * sa, cb * ca}};
        double N[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c) N[r * 4 + c] = R[r][0] * T[0 * 4 + c] + R[r][1] * T[1 * 4 + c] + R[r][2] * T[2 * 4 + c];
            N[r * 4 + 3] += x[3 + r];
        }
        memcpy(T, N, sizeof(T));
    }
    (void)hipFree(ds); (void)hipFree(dt); (void)hipFree(dn); (void)hipFree(dT); (void)hipFree(dP);
    if (rc) { set_error("stocs_icp_point_to_plane: HIP error"); return rc; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) T16_out[c * 4 + r] = (float)T[r * 4 + c];
    T16_out[3] = T16_out[7] = T16_out[11] = 0.f; T16_out[15] = 1.f;
    if (n_correspondences) *n_correspondences = ncorr;
    return STOCS_OK;
}
