// ingest.hip -- the two steps immediately UPSTREAM of the hot path, on the GPU:
//   scene ingest          rgbd::load_rgbd_data_sampled          reference src/rgbd.cpp:179-281
//   model preprocessing   stocs::pre_process_model (cloud part)  reference src/stocs.cpp:28-60
// The reference delegates their arithmetic to PCL (VoxelGrid, RadiusOutlierRemoval, NormalEstimation)
// and OpenCV-contrib (RgbdNormals LINEMOD), none of which is available here: parity with the
// reference is UNPINNED for these rows.  They are pinned instead against this repo's numpy restatement
// (oracle/ingest_oracle.py, the script that produced tests/golden/example_*.npz): identical voxel
// membership, outlier decisions, pixels and probabilities; centroids and normals to float rounding.
//
// Kernels: per-pixel back-projection and 5x5 plane-fit normals (double-precision covariance + cyclic
// Jacobi), voxel grid = radix sort of leaf indices + segmented centroid, radius outlier removal =
// uniform-grid neighbour count, stable compaction of the survivors.  All integer/byte-heavy and
// HBM/L2-bound; no MFMA.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <cstring>


#include <limits.h>

#include <algorithm>
#include <map>
#include <vector>

#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

struct D3 { double x, y, z; };

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi in double); returns eigenvalue
__device__ double smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double v[3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double offd = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (offd < 1e-300) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            if (A[p][q] == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
        }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    v[0] = V[0][m]; v[1] = V[1][m]; v[2] = V[2][m];
    return A[m][m];
}

// rgbd.cpp:207-225: every pixel becomes a point (zero depth -> the origin)
__global__ __launch_bounds__(256) void backproject_kernel(const uint16_t* __restrict__ depth, int W, int H, float fx, float cx, float fy, float cy,
                                                          float depth_scale, float4* __restrict__ P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W * H) return;
    const int i = idx / W, j = idx % W;
    const float d = (float)depth[idx] * depth_scale;
    P[idx] = make_float4((float)(((double)j - (double)cx) * (double)d / (double)fx), (float)(((double)i - (double)cy) * (double)d / (double)fy), d, 0.f);
}

// STOCS_NORMALS_PLANE_FIT (the stand-in of rounds 1-2 for rgbd.cpp:203, kept for the fixtures it produced and as an A/B):
// least-squares plane over the valid pixels of the 5x5 window, oriented toward the camera; NaN where unreliable
__global__ __launch_bounds__(256) void depth_normals_kernel(const float4* __restrict__ P, int W, int H, float4* __restrict__ N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W * H) return;
    const int i = idx / W, j = idx % W;
    const float nanf_ = __int_as_float(0x7fc00000);
    float4 out = make_float4(nanf_, nanf_, nanf_, 0.f);
    const float4 pc = P[idx];
    if (pc.z > 0.f) {
        double n = 0, sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
        for (int di = -2; di <= 2; ++di)
            for (int dj = -2; dj <= 2; ++dj) {
                const int ii = i + di, jj = j + dj;
                if (ii < 0 || jj < 0 || ii >= H || jj >= W) continue;
                const float4 p = P[ii * W + jj];
                if (!(p.z > 0.f)) continue;
                const double x = p.x, y = p.y, z = p.z;
                n += 1; sx += x; sy += y; sz += z;
                sxx += x * x; sxy += x * y; sxz += x * z; syy += y * y; syz += y * z; szz += z * z;
            }
        if (n >= 6) {
            const double mx = sx / n, my = sy / n, mz = sz / n;
            double v[3];
            const double w0 = smallest_eigvec(sxx / n - mx * mx, sxy / n - mx * my, sxz / n - mx * mz, syy / n - my * my, syz / n - my * mz, szz / n - mz * mz, v);
            if (!(w0 > 1e-5)) {  // plane residual variance <= (3.2 mm)^2
                double s = (v[0] * pc.x + v[1] * pc.y + v[2] * pc.z) > 0 ? -1.0 : 1.0;  // toward the camera: n . p < 0
                out = make_float4((float)(s * v[0]), (float)(s * v[1]), (float)(s * v[2]), 0.f);
            }
        }
    }
    N[idx] = out;
}

// cv::rgbd::RgbdNormals(rows, cols, CV_32F, K, 5, RGBD_NORMALS_METHOD_LINEMOD) applied to the raw 16-bit depth image
// (rgbd.cpp:199-205), restated from the published method -- Hinterstoisser et al., PAMI 2012, section 2.4: the depth gradient
// that best explains, in the least-squares sense, the depth differences to the 8 neighbours at +-5 pixels (neighbours whose depth
// differs from the centre by 50 raw units or more are left out: they lie across a depth edge); the normal is that of the plane
// through the back-projected points X, X1 = v(x+1, y)(D + dD/dx), X2 = v(x, y+1)(D + dD/dy).  Sums and the 2x2 solve in 64-bit
// integers (the division by the determinant is dropped: it scales both tangent vectors), the cross product in float, the result
// normalised and pointed at the camera (z <= 0).  Pixels within 5 (6 at the far sides) of the border, and pixels whose patch
// gives a zero vector (no admissible neighbour, zero depth), stay NaN: rgbd.cpp:263-267 drops them.  OpenCV is absent: UNPINNED.
__global__ __launch_bounds__(256) void gradient_normals_kernel(const uint16_t* __restrict__ depth, int W, int H, float fx, float cx, float fy, float cy,
                                                               float4* __restrict__ N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W * H) return;
    const int y = idx / W, x = idx % W;
    const float nanf_ = __int_as_float(0x7fc00000);
    float4 out = make_float4(nanf_, nanf_, nanf_, 0.f);
    const int r = 5;
    if (y >= r && y < H - r - 1 && x >= r && x < W - r - 1) {
        const long long d = depth[idx];
        long long A0 = 0, A1 = 0, A3 = 0, b0 = 0, b1 = 0;
        for (int j = -r; j <= r; j += r)
            for (int i = -r; i <= r; i += r) {
                const long long delta = (long long)depth[(y + j) * W + (x + i)] - d;
                if (delta >= 50 || delta <= -50) continue;   // strict: |delta| < 50 is kept (the library's accumBilateral test as recalled; UNPINNED, OpenCV is absent)
                A0 += i * i; A1 += i * j; A3 += j * j;
                b0 += i * delta; b1 += j * delta;
            }
        const long long det = A0 * A3 - A1 * A1;
        const long long gx = A3 * b0 - A1 * b1, gy = -A1 * b0 + A0 * b1;   // det * (dD/dx, dD/dy)
        // K^-1 by hand: (1/fx, 0, -cx/fx; 0, 1/fy, -cy/fy; 0, 0, 1), in float
        const float k00 = 1.0f / fx, k02 = (0.0f * cy - cx * fy) / (fx * fy), k11 = 1.0f / fy, k12 = -cy / fy;
        const float a1 = (float)(d * det + (long long)(x + 1) * gx), b1f = (float)((long long)y * gx), c1 = (float)gx;
        const float a2 = (float)((long long)x * gy), b2f = (float)(d * det + (long long)(y + 1) * gy), c2 = (float)gy;
        const float X1x = k00 * a1 + (0.0f * b1f + k02 * c1), X1y = k11 * b1f + k12 * c1, X1z = c1;
        const float X2x = k00 * a2 + (0.0f * b2f + k02 * c2), X2y = k11 * b2f + k12 * c2, X2z = c2;
        const float nx = X1y * X2z - X1z * X2y, ny = X1z * X2x - X1x * X2z, nz = X1x * X2y - X1y * X2x;
        const double len = sqrt((double)nx * nx + (double)ny * ny + (double)nz * nz);
        if (len > 0) {
            const double s = (nz > 0 ? -1.0 : 1.0) / len;
            out = make_float4((float)(nx * s), (float)(ny * s), (float)(nz * s), 0.f);
        }
    }
    N[idx] = out;
}

// pcl::VoxelGrid leaf coordinates floor(p / leaf) (relative to the minimum added on the host)
__global__ __launch_bounds__(256) void leaf_coords_kernel(const float4* __restrict__ P, int n, double inv_leaf, int3* __restrict__ ijk, int* __restrict__ mm) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 6) mm[threadIdx.x] = threadIdx.x < 3 ? INT_MAX : INT_MIN;   // for minmax_i3_kernel, which runs after this one
    if (idx >= n) return;
    const float4 p = P[idx];
    ijk[idx] = make_int3((int)floor((double)p.x * inv_leaf), (int)floor((double)p.y * inv_leaf), (int)floor((double)p.z * inv_leaf));
}
// KeyT = uint32_t when the leaf grid has fewer than 2^32 cells (any camera frame): half the bytes through the sort
template <class KeyT>
__global__ __launch_bounds__(256) void leaf_keys_kernel(const int3* __restrict__ ijk, int n, int3 mn, int3 dims, KeyT* __restrict__ keys,
                                                        uint32_t* __restrict__ ids) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int3 c = ijk[idx];
    keys[idx] = (KeyT)((uint64_t)(c.x - mn.x) + (uint64_t)(c.y - mn.y) * (uint64_t)dims.x + (uint64_t)(c.z - mn.z) * (uint64_t)dims.x * (uint64_t)dims.y);
    ids[idx] = (uint32_t)idx;
}
template <class KeyT>
__global__ __launch_bounds__(256) void seg_heads_kernel(const KeyT* __restrict__ keys, int n, uint32_t* __restrict__ head) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    head[idx] = (idx == 0 || keys[idx] != keys[idx - 1]) ? 1u : 0u;
}
// first sorted position of every leaf (+ the end sentinel)
__global__ __launch_bounds__(256) void seg_start_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ seg_of, int n, int nv,
                                                        uint32_t* __restrict__ seg_start) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (head[idx]) seg_start[seg_of[idx]] = (uint32_t)idx;
    if (idx == 0) seg_start[nv] = (uint32_t)n;
}
// Leaf centroids in double.  One thread per leaf sums a short leaf in sorted (= original, the sort is stable) order;
// a long leaf -- e.g. the one that collects every zero-depth pixel at the origin, > 10^5 points -- is queued for
// centroid_long_kernel instead of serialising one thread for milliseconds.
#define STOCS_LONG_LEAF 192
__global__ __launch_bounds__(256) void centroid_kernel(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ ids, int nv,
                                                       const float4* __restrict__ P, const float4* __restrict__ extra, float4* __restrict__ cen,
                                                       float4* __restrict__ ext, uint32_t* __restrict__ long_list, uint32_t* __restrict__ n_long,
                                                       double* __restrict__ long_acc, uint32_t* __restrict__ long_done) {
    // eight lanes per leaf (a leaf holds ~15 points of a frame; one thread per leaf walked them as a chain of dependent gathers):
    // strided partial sums in double, then a fixed xor tree -- the same value for every run, and the sequential sum's
    // whenever that is exact (see centroid_long_kernel)
    const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const bool live = s < nv;
    uint32_t e0 = 0, e1 = 0;
    if (live) { e0 = seg_start[s]; e1 = seg_start[s + 1]; }
    const bool is_long = live && e1 - e0 > STOCS_LONG_LEAF;
    if (is_long && sub == 0) {
        const uint32_t j = atomicAdd(n_long, 1u);
        long_list[j] = (uint32_t)s;
        long_done[j] = 0;
    }
    double sx = 0, sy = 0, sz = 0, ex = 0, ey = 0, ez = 0;
    if (live && !is_long)
        for (uint32_t e = e0 + sub; e < e1; e += 8) {
            const float4 p = P[ids[e]];
            sx += p.x; sy += p.y; sz += p.z;
            if (extra) { const float4 q = extra[ids[e]]; ex += q.x; ey += q.y; ez += q.z; }
        }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); sz += __shfl_xor(sz, off, 64);
        if (extra) { ex += __shfl_xor(ex, off, 64); ey += __shfl_xor(ey, off, 64); ez += __shfl_xor(ez, off, 64); }
    }
    if (!live || is_long || sub != 0) return;
    const double cnt = (double)(e1 - e0);
    cen[s] = make_float4((float)(sx / cnt), (float)(sy / cnt), (float)(sz / cnt), 0.f);
    if (extra) ext[s] = make_float4((float)(ex / cnt), (float)(ey / cnt), (float)(ez / cnt), 0.f);
}
// Queued leaves (the leaf of the invalid pixels at the origin holds a third of a frame): every workgroup takes a strided
// share of every queued leaf and writes its partial sums; the workgroup that arrives last adds the partial sums in
// workgroup order and writes the centroid -- a fixed summation tree whatever the arrival order (deterministic), and equal
// to the sequential sum whenever that is exact in double (points of one 5 mm leaf: always, short of denormal coordinates).
#define STOCS_LONG_GROUPS 64
__global__ __launch_bounds__(256) void centroid_long_kernel(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ ids,
                                                            const float4* __restrict__ P, const float4* __restrict__ extra, float4* __restrict__ cen,
                                                            float4* __restrict__ ext, const uint32_t* __restrict__ long_list,
                                                            const uint32_t* __restrict__ n_long, double* __restrict__ long_acc,
                                                            uint32_t* __restrict__ long_done) {
    __shared__ double sh[6][4];
    __shared__ double sh_part[6 * STOCS_LONG_GROUPS];
    __shared__ uint32_t sh_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t j = 0; j < *n_long; ++j) {
        const uint32_t s = long_list[j];
        const uint32_t e0 = seg_start[s], e1 = seg_start[s + 1];
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t e = e0 + blockIdx.x * 256 + threadIdx.x; e < e1; e += gridDim.x * 256) {
            const float4 p = P[ids[e]];
            v[0] += p.x; v[1] += p.y; v[2] += p.z;
            if (extra) { const float4 q = extra[ids[e]]; v[3] += q.x; v[4] += q.y; v[5] += q.z; }
        }
        for (int k = 0; k < 6; ++k) {
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
            if (lane == 0) sh[k][wv] = v[k];
        }
        __syncthreads();
        if (threadIdx.x < 6)
            __hip_atomic_store(&long_acc[(6 * (size_t)j + threadIdx.x) * STOCS_LONG_GROUPS + blockIdx.x],
                               (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) sh_last = atomicAdd(&long_done[j], 1u) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (sh_last) {   // (uniform) the partial sums arrive through LDS -- 6 x 64 loads by as many threads at once -- and are added in workgroup order as before
            __threadfence();
            for (unsigned i = threadIdx.x; i < 6u * gridDim.x; i += 256u)
                sh_part[i] = __hip_atomic_load(&long_acc[(6 * (size_t)j + i / gridDim.x) * STOCS_LONG_GROUPS + i % gridDim.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (threadIdx.x == 0) {
                const double cnt = (double)(e1 - e0);
                double a[6];
                for (int k = 0; k < 6; ++k) {
                    a[k] = 0.0;
                    for (unsigned g = 0; g < gridDim.x; ++g) a[k] += sh_part[(unsigned)k * gridDim.x + g];
                }
                cen[s] = make_float4((float)(a[0] / cnt), (float)(a[1] / cnt), (float)(a[2] / cnt), 0.f);
                if (extra) ext[s] = make_float4((float)(a[3] / cnt), (float)(a[4] / cnt), (float)(a[5] / cnt), 0.f);
            }
        }
        __syncthreads();
    }
}

// pcl::RadiusOutlierRemoval (rgbd.cpp:233-237): number of points (itself included) within radius
__global__ __launch_bounds__(256) void ror_cell_kernel(const float4* __restrict__ P, int n, double3 mn, double inv_r, int3 dims, uint32_t* __restrict__ cell) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 p = P[idx];
    const int cx = (int)floor(((double)p.x - mn.x) * inv_r), cy = (int)floor(((double)p.y - mn.y) * inv_r), cz = (int)floor(((double)p.z - mn.z) * inv_r);
    cell[idx] = (uint32_t)((cz * dims.y + cy) * dims.x + cx);
}
__global__ __launch_bounds__(256) void cell_start_kernel(const uint32_t* __restrict__ sorted_cell, int n, uint32_t* __restrict__ start, uint32_t* __restrict__ end) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t c = sorted_cell[idx];
    if (idx == 0 || sorted_cell[idx - 1] != c) start[c] = (uint32_t)idx;
    if (idx == n - 1 || sorted_cell[idx + 1] != c) end[c] = (uint32_t)idx + 1;
}
// (only "more than min_pts" is ever asked of the count -- scene_select_kernel -- so a point stops counting there: a surface point has ~100
//  neighbours within the radius and was walking all of them, one dependent gather after the other: 74 us of a frame's ingest in round 5a)
__global__ __launch_bounds__(256) void ror_count_kernel(const float4* __restrict__ P, int n, double3 mn, double inv_r, int3 dims, double radius,
                                                        const uint32_t* __restrict__ start, const uint32_t* __restrict__ end,
                                                        const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ count, uint32_t min_pts) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 p = P[idx];
    const int cx = (int)floor(((double)p.x - mn.x) * inv_r), cy = (int)floor(((double)p.y - mn.y) * inv_r), cz = (int)floor(((double)p.z - mn.z) * inv_r);
    uint32_t k = 0;
    const double r2 = radius * radius;
    for (int dz = -1; dz <= 1 && k <= min_pts; ++dz)                  // (the test once per slab of nine cells: their look-ups stay independent)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x < 0 || y < 0 || z < 0 || x >= dims.x || y >= dims.y || z >= dims.z) continue;
                const uint32_t c = (uint32_t)((z * dims.y + y) * dims.x + x);
                const uint32_t e1 = end[c];
                for (uint32_t e = start[c]; e < e1; ++e) {               // (a cell's run is walked whole: its gathers stay independent of the count)
                    const float4 q = P[sorted_ids[e]];
                    const double ddx = (double)p.x - q.x, ddy = (double)p.y - q.y, ddz = (double)p.z - q.z;
                    if (ddx * ddx + ddy * ddy + ddz * ddz <= r2) k++;
                }
            }
    count[idx] = k;   // (a lower bound beyond min_pts)
}

// rgbd.cpp:240-278: z range, re-projection to the pixel, class probability threshold, normal validity
__global__ __launch_bounds__(256) void scene_select_kernel(const float4* __restrict__ cen, const uint32_t* __restrict__ count, int n, uint32_t min_pts,
                                                           float fx, float cx, float fy, float cy, int W, int H, const uint16_t* __restrict__ prob,
                                                           float class_threshold, const float4* __restrict__ normals, uint32_t* __restrict__ keep,
                                                           float4* __restrict__ out_n, float* __restrict__ out_p, int2* __restrict__ out_px) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint32_t k = 0;
    const float4 pt = cen[idx];
    if (count[idx] > min_pts && pt.z == pt.z && pt.z > 0.f && pt.z <= 2.0f) {
        const int col = (int)((fx * pt.x + cx * pt.z) / pt.z);
        const int row = (int)((fy * pt.y + cy * pt.z) / pt.z);
        if (row >= 0 && row < H && col >= 0 && col < W) {
            const float cp = (float)((double)prob[row * W + col] * (1.0 / 10000));
            const float4 nn = normals[row * W + col];
            const bool fin = nn.x == nn.x && nn.y == nn.y && nn.z == nn.z;
            if (!(cp < class_threshold) && fin && !(nn.x == 0 && nn.y == 0 && nn.z == 0)) {
                k = 1;
                out_n[idx] = nn; out_p[idx] = cp; out_px[idx] = make_int2(row, col);
            }
        }
    }
    keep[idx] = k;
}

// model normals: pcl::NormalEstimation with a radius search (rgbd.cpp:72-83), flipped toward the origin
// and negated (stocs.cpp:47-52) => pointing away from the model origin.  Brute-force neighbour scan.
__global__ __launch_bounds__(256) void model_normals_kernel(const float4* __restrict__ P, int n, double radius, float4* __restrict__ N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 pc = P[idx];
    const double r2 = radius * radius;
    double cnt = 0, sx = 0, sy = 0, sz = 0;
    for (int j = 0; j < n; ++j) {
        const float4 q = P[j];
        const double dx = (double)q.x - pc.x, dy = (double)q.y - pc.y, dz = (double)q.z - pc.z;
        if (dx * dx + dy * dy + dz * dz <= r2) { cnt += 1; sx += q.x; sy += q.y; sz += q.z; }
    }
    const float nanf_ = __int_as_float(0x7fc00000);
    float4 out = make_float4(nanf_, nanf_, nanf_, 0.f);
    if (cnt >= 3) {
        const double mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
        double cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
        for (int j = 0; j < n; ++j) {
            const float4 q = P[j];
            const double dx = (double)q.x - pc.x, dy = (double)q.y - pc.y, dz = (double)q.z - pc.z;
            if (dx * dx + dy * dy + dz * dz <= r2) {
                const double x = q.x - mx, y = q.y - my, z = q.z - mz;
                cxx += x * x; cxy += x * y; cxz += x * z; cyy += y * y; cyz += y * z; czz += z * z;
            }
        }
        double v[3];
        smallest_eigvec(cxx, cxy, cxz, cyy, cyz, czz, v);
        double s = (v[0] * -pc.x + v[1] * -pc.y + v[2] * -pc.z) < 0 ? -1.0 : 1.0;  // flipNormalTowardsViewpoint(0,0,0)
        s = -s;                                                                       // stocs.cpp:47-52
        out = make_float4((float)(s * v[0]), (float)(s * v[1]), (float)(s * v[2]), 0.f);
    }
    N[idx] = out;
}

// ---- workspace: every temporary of a call comes out of a grow-only arena cached per host thread and device, so a
// caller that ingests one frame after the other does no hipMalloc / hipFree after the first frame.
// stocs_trim() gives the calling thread's cached memory back.
static thread_local std::map<int, Arena>* tl_ws = NULL;
static thread_local Arena* tl_cur = NULL;

// Pinned staging of a frame's images (in) and cloud (out), per calling thread: a copy between device memory and the caller's PAGEABLE arrays
// goes through the runtime's own staging path -- the four output copies of a frame stalled the calling thread for 240-280 us (kernel trace of
// tools/frame_latency.py, round 5b); through this block they are device -> pinned copies and host memcpy's of half a megabyte.
static thread_local void* tl_pin = NULL;
static thread_local size_t tl_pin_bytes = 0;
static int staging(size_t bytes, char** out) {
    if (tl_pin_bytes < bytes) {
        if (tl_pin) { (void)hipHostFree(tl_pin); tl_pin = NULL; tl_pin_bytes = 0; }
        const size_t want = bytes + bytes / 4 + ((size_t)1 << 20);
        STOCS_HIP_CHECK(pinned_malloc(&tl_pin, want));
        tl_pin_bytes = want;
    }
    *out = (char*)tl_pin;
    return STOCS_OK;
}

static int workspace_begin(int device) {
    if (!tl_ws) tl_ws = new std::map<int, Arena>();
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    tl_cur = &(*tl_ws)[dev];
    return tl_cur->reset();   // the previous call synchronised before it returned
}

template <class T>
struct Buf {   // typed view of workspace memory
    T* p;
    Buf() : p(NULL) {}
    int alloc(size_t n) { return tl_cur->take(n * sizeof(T), (void**)&p); }
};

// out[0..2] = min, out[3..5] = max of the leaf coordinates (initialised by leaf_coords_kernel): strided partial results,
// wavefront reductions, one atomic per wavefront and component
__global__ __launch_bounds__(256) void minmax_i3_kernel(const int3* __restrict__ v, int n, int* __restrict__ out) {
    // (one atomic per WORKGROUP and component: 256 workgroups x 4 wavefronts x 6 atomics on six addresses serialised in the L2 -- 73 us of a
    //  frame's ingest in round 5a for a 300 000-point reduction)
    __shared__ int s_m[6][4];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int3 c = v[i];
        mn[0] = min(mn[0], c.x); mn[1] = min(mn[1], c.y); mn[2] = min(mn[2], c.z);
        mx[0] = max(mx[0], c.x); mx[1] = max(mx[1], c.y); mx[2] = max(mx[2], c.z);
    }
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off > 0; off >>= 1) { mn[k] = min(mn[k], __shfl_xor(mn[k], off, 64)); mx[k] = max(mx[k], __shfl_xor(mx[k], off, 64)); }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { s_m[k][threadIdx.x >> 6] = mn[k]; s_m[3 + k][threadIdx.x >> 6] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        if (k < 3) atomicMin(&out[k], min(min(s_m[k][0], s_m[k][1]), min(s_m[k][2], s_m[k][3])));
        else atomicMax(&out[k], max(max(s_m[k][0], s_m[k][1]), max(s_m[k][2], s_m[k][3])));
    }
}
__global__ __launch_bounds__(256) void iota_kernel(uint32_t* __restrict__ a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void zero_u32x2_kernel(uint32_t* __restrict__ a, uint32_t* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0; b[i] = 0; }
}
// survivors, in voxel order (stable): pos | nrm | prob | pixel packed for one copy per field
__global__ __launch_bounds__(256) void scene_pack_kernel(const float4* __restrict__ cen, const float4* __restrict__ nn, const float* __restrict__ pp,
                                                         const int2* __restrict__ px, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                                                         int n, float* __restrict__ o_pos, float* __restrict__ o_nrm, float* __restrict__ o_prob,
                                                         int32_t* __restrict__ o_px) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const uint32_t m = pos[i];
    o_pos[3 * m] = cen[i].x; o_pos[3 * m + 1] = cen[i].y; o_pos[3 * m + 2] = cen[i].z;
    o_nrm[3 * m] = nn[i].x; o_nrm[3 * m + 1] = nn[i].y; o_nrm[3 * m + 2] = nn[i].z;
    o_prob[m] = pp[i];
    o_px[2 * m] = px[i].x; o_px[2 * m + 1] = px[i].y;
}

// voxel grid on device points dP[n] (+ optional extra field); outputs device centroid arrays (workspace memory)
static int voxel_grid_device(const float4* dP, const float4* dExtra, int n, double leaf, Buf<float4>& cen, Buf<float4>& ext, int* n_out, hipStream_t st,
                             double* box6 = NULL, int* pin_small = NULL) {   // pin_small: 16 pinned words for the call's small read-backs (else pageable stack words)   // box6: a box that holds every centroid (from the leaf bounds, a leaf of slack either side)
    *n_out = 0;
    if (n == 0) return STOCS_OK;
    Buf<int3> ijk; Buf<uint64_t> keys, keys_s; Buf<uint32_t> ids, ids_s, head, seg; Buf<char> tmp; Buf<int> mm;
    int rc;
    if ((rc = ijk.alloc(n)) || (rc = keys.alloc(n)) || (rc = keys_s.alloc(n)) || (rc = ids.alloc(n)) || (rc = ids_s.alloc(n)) || (rc = head.alloc(n)) ||
        (rc = seg.alloc(n)) || (rc = mm.alloc(8)))
        return rc;
    const dim3 g((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(leaf_coords_kernel, g, dim3(256), 0, st, dP, n, 1.0 / leaf, ijk.p, mm.p);
    hipLaunchKernelGGL(minmax_i3_kernel, dim3(std::min<unsigned>(g.x, 128u)), dim3(256), 0, st, ijk.p, n, mm.p);
    int h6_stack[6];
    int* h6 = pin_small ? pin_small : h6_stack;
    STOCS_HIP_CHECK(hipMemcpyAsync(h6, mm.p, sizeof(h6_stack), hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    const int3 mn = make_int3(h6[0], h6[1], h6[2]), mx = make_int3(h6[3], h6[4], h6[5]);
    if (box6) for (int k = 0; k < 3; ++k) { box6[k] = ((double)h6[k] - 1.0) * leaf; box6[3 + k] = ((double)h6[3 + k] + 2.0) * leaf; }
    const int3 dims = make_int3(mx.x - mn.x + 1, mx.y - mn.y + 1, mx.z - mn.z + 1);
    if ((double)dims.x * dims.y * dims.z > 9.0e18) { set_error("voxel grid: leaf size too small for the cloud extent"); return STOCS_ERR_INVALID; }
    int key_bits = 1;   // sort only the bits the keys can have
    while (key_bits < 64 && (double)(1ull << key_bits) < (double)dims.x * dims.y * dims.z) key_bits++;
    size_t tb = 0, tb2 = 0;
    bool own_sort = false;
    const uint32_t* own_err = NULL;
    char* scan_tmp = NULL;
    STOCS_HIP_CHECK(exclusive_scan(NULL, tb2, head.p, seg.p, (size_t)n, st));
    if (key_bits <= 32) {
        uint32_t* k32 = (uint32_t*)keys.p; uint32_t* k32s = (uint32_t*)keys_s.p;
        hipLaunchKernelGGL(leaf_keys_kernel<uint32_t>, g, dim3(256), 0, st, ijk.p, n, mn, dims, k32, ids.p);
        // a frame's 300 000 leaf keys: the library's own onesweep (sort32.hip, one launch per 8-bit pass) where rocPRIM's radix_sort_pairs runs
        // ~20 small launches (190 us of a frame's ingest in round 5a); small clouds stay with rocPRIM (a block / merge sort there).  Both stable.
        own_sort = n >= 65536 && !(getenv("STOCS_SORT") && !strcmp(getenv("STOCS_SORT"), "rocprim"));
        Buf<char> tmp_scan;
        if (own_sort) {
            STOCS_HIP_CHECK(sort_pairs_own(NULL, tb, k32, k32s, ids.p, ids_s.p, (size_t)n, 0, (unsigned)key_bits, NULL, 1, st));
            if ((rc = tmp.alloc(tb)) || (rc = tmp_scan.alloc(tb2))) return rc;   // (the sort's error word is read after the scan: separate blocks)
            STOCS_HIP_CHECK(sort_pairs_own(tmp.p, tb, k32, k32s, ids.p, ids_s.p, (size_t)n, 0, (unsigned)key_bits, NULL, 1, st));
            own_err = (const uint32_t*)(tmp.p + sort_own_err_offset());
            scan_tmp = tmp_scan.p;
        } else {
        STOCS_HIP_CHECK(sort_pairs(NULL, tb, k32, k32s, ids.p, ids_s.p, (size_t)n, 0, (unsigned)key_bits, st));   // stable
        if ((rc = tmp.alloc(std::max(tb, tb2)))) return rc;
        STOCS_HIP_CHECK(sort_pairs(tmp.p, tb, k32, k32s, ids.p, ids_s.p, (size_t)n, 0, (unsigned)key_bits, st));
        scan_tmp = tmp.p;
        }
        hipLaunchKernelGGL(seg_heads_kernel<uint32_t>, g, dim3(256), 0, st, k32s, n, head.p);
    } else {
        hipLaunchKernelGGL(leaf_keys_kernel<uint64_t>, g, dim3(256), 0, st, ijk.p, n, mn, dims, keys.p, ids.p);
        STOCS_HIP_CHECK(sort_pairs(NULL, tb, keys.p, keys_s.p, ids.p, ids_s.p, (size_t)n, 0, (unsigned)key_bits, st));   // stable
        if ((rc = tmp.alloc(std::max(tb, tb2)))) return rc;
        STOCS_HIP_CHECK(sort_pairs(tmp.p, tb, keys.p, keys_s.p, ids.p, ids_s.p, (size_t)n, 0, (unsigned)key_bits, st));
        scan_tmp = tmp.p;
        hipLaunchKernelGGL(seg_heads_kernel<uint64_t>, g, dim3(256), 0, st, keys_s.p, n, head.p);
    }
    STOCS_HIP_CHECK(exclusive_scan(scan_tmp, tb2, head.p, seg.p, (size_t)n, st));
    uint32_t rb_stack[3] = {0, 0, 0};
    uint32_t* rb = pin_small ? (uint32_t*)pin_small + 8 : rb_stack;
    rb[0] = rb[1] = rb[2] = 0u;
    STOCS_HIP_CHECK(hipMemcpyAsync(&rb[0], seg.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(&rb[1], head.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
    if (own_err) STOCS_HIP_CHECK(hipMemcpyAsync(&rb[2], own_err, 4, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t last_seg = rb[0], last_head = rb[1], sort_err = rb[2];
    if (sort_err) { set_error("voxel grid: the leaf sort gave up waiting for a tile (sort32.hip)"); return STOCS_ERR_HIP; }
    const int nv = (int)(last_seg + last_head);
    Buf<uint32_t> seg_start, long_list, n_long, long_done;
    Buf<double> long_acc;
    const size_t max_long = (size_t)n / STOCS_LONG_LEAF + 1;   // leaves with more than STOCS_LONG_LEAF points each
    if ((rc = cen.alloc(nv)) || (rc = ext.alloc(nv)) || (rc = seg_start.alloc((size_t)nv + 1)) || (rc = long_list.alloc(nv)) || (rc = n_long.alloc(2)) ||
        (rc = long_acc.alloc(6 * STOCS_LONG_GROUPS * max_long)) || (rc = long_done.alloc(max_long)))
        return rc;
    hipLaunchKernelGGL(seg_start_kernel, g, dim3(256), 0, st, head.p, seg.p, n, nv, seg_start.p);
    hipLaunchKernelGGL(zero_u32x2_kernel, dim3(1), dim3(256), 0, st, n_long.p, n_long.p + 1, (size_t)1);
    hipLaunchKernelGGL(centroid_kernel, dim3((unsigned)(((size_t)nv * 8 + 255) / 256)), dim3(256), 0, st, seg_start.p, ids_s.p, nv, dP, dExtra, cen.p, ext.p, long_list.p,
                       n_long.p, long_acc.p, long_done.p);
    hipLaunchKernelGGL(centroid_long_kernel, dim3(STOCS_LONG_GROUPS), dim3(256), 0, st, seg_start.p, ids_s.p, dP, dExtra, cen.p, ext.p, long_list.p, n_long.p, long_acc.p,
                       long_done.p);
    STOCS_HIP_CHECK(hipGetLastError());
    *n_out = nv;
    return STOCS_OK;
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_trim(void) {
    if (tl_ws) {
        (void)hipDeviceSynchronize();
        for (std::map<int, Arena>::iterator it = tl_ws->begin(); it != tl_ws->end(); ++it) it->second.destroy();
        delete tl_ws;
        tl_ws = NULL;
        tl_cur = NULL;
    }
    if (tl_pin) { (void)hipHostFree(tl_pin); tl_pin = NULL; tl_pin_bytes = 0; }
    return STOCS_OK;
}

int stocs_ingest_scene(const stocs_camera* cam, const uint16_t* depth, const uint16_t* class_prob, float voxel_size,
                       float class_threshold, int device, float* pos3, float* nrm3, float* prob, int32_t* pixel2, int cap, int* n_out) {
    if (!cam || !depth || !class_prob || !n_out || cam->width <= 0 || cam->height <= 0 || !(voxel_size > 0)) return STOCS_ERR_INVALID;
    if (cam->normal_method != STOCS_NORMALS_DEPTH_GRADIENT && cam->normal_method != STOCS_NORMALS_PLANE_FIT) { set_error("stocs_ingest_scene: unknown normal_method %d", cam->normal_method); return STOCS_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available: this library has no CPU fallback"); return STOCS_ERR_NO_DEVICE; }
    if (device >= 0) STOCS_HIP_CHECK(hipSetDevice(device));
    const bool dbg = getenv("STOCS_DEBUG_TIMING") != NULL;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto tick = [&](const char* label) {
        if (!dbg) return;
        (void)hipDeviceSynchronize();
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "[stocs ingest] %-22s %8.3f ms\n", label, (t1.tv_sec - ts0.tv_sec) * 1e3 + (t1.tv_nsec - ts0.tv_nsec) * 1e-6);
        ts0 = t1;
    };
    int rc = workspace_begin(device);
    if (rc) return rc;
    tick("workspace");
    hipStream_t st = NULL;
    const int W = cam->width, H = cam->height, npx = W * H;
    Buf<uint16_t> dD, dC; Buf<float4> dP, dN;
    if ((rc = dD.alloc(npx)) || (rc = dC.alloc(npx)) || (rc = dP.alloc(npx)) || (rc = dN.alloc(npx))) return rc;
    char* pin = NULL;
    if ((rc = staging((size_t)npx * 40 + 4096, &pin))) return rc;     // in: two u16 images (4 B per pixel); out: at most one point per pixel, 36 B each
    memcpy(pin, depth, 2 * (size_t)npx); memcpy(pin + 2 * (size_t)npx, class_prob, 2 * (size_t)npx);
    STOCS_HIP_CHECK(hipMemcpyAsync(dD.p, pin, 2 * (size_t)npx, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(dC.p, pin + 2 * (size_t)npx, 2 * (size_t)npx, hipMemcpyHostToDevice, st));
    const dim3 g((unsigned)((npx + 255) / 256));
    hipLaunchKernelGGL(backproject_kernel, g, dim3(256), 0, st, dD.p, W, H, cam->fx, cam->cx, cam->fy, cam->cy, cam->depth_scale, dP.p);
    if (cam->normal_method == STOCS_NORMALS_PLANE_FIT) hipLaunchKernelGGL(depth_normals_kernel, g, dim3(256), 0, st, dP.p, W, H, dN.p);
    else hipLaunchKernelGGL(gradient_normals_kernel, g, dim3(256), 0, st, dD.p, W, H, cam->fx, cam->cx, cam->fy, cam->cy, dN.p);
    tick("upload+backproject+normals");
    Buf<float4> cen, ext;
    int nv = 0;
    double box6[6] = {0, 0, 0, 0, 0, 0};
    int* pin_small = (int*)(pin + (size_t)npx * 40);                 // (the 4 KB behind the cloud's part of the staging block)
    if ((rc = voxel_grid_device(dP.p, NULL, npx, (double)voxel_size, cen, ext, &nv, st, box6, pin_small))) return rc;   // rgbd.cpp:228-231
    *n_out = 0;
    tick("voxel grid");
    if (nv == 0) { STOCS_HIP_CHECK(hipStreamSynchronize(st)); return STOCS_OK; }
    // radius outlier removal: radius 2*voxel + 5 mm, more than 10 points (itself included)   rgbd.cpp:233-237
    const double radius = 2.0 * (double)voxel_size + 0.005;
    // the search grid only has to hold every centroid (the counts do not depend on where its cells fall): its box comes from
    // the leaf bounds the voxel grid already read back -- no reduction over the centroids, no synchronisation here
    const double3 mn = make_double3(box6[0], box6[1], box6[2]), mx = make_double3(box6[3], box6[4], box6[5]);
    const int3 dims = make_int3((int)floor((mx.x - mn.x) / radius) + 1, (int)floor((mx.y - mn.y) / radius) + 1, (int)floor((mx.z - mn.z) / radius) + 1);
    const size_t ncell = (size_t)dims.x * dims.y * dims.z;
    if (ncell > ((size_t)1 << 28)) { set_error("scene extent too large for the outlier-removal grid"); return STOCS_ERR_INVALID; }
    Buf<uint32_t> cell, cell_s, ids, ids_s, cstart, cend, count, keep, kpos; Buf<char> tmp;
    if ((rc = cell.alloc(nv)) || (rc = cell_s.alloc(nv)) || (rc = ids.alloc(nv)) || (rc = ids_s.alloc(nv)) || (rc = cstart.alloc(ncell)) || (rc = cend.alloc(ncell)) ||
        (rc = count.alloc(nv)) || (rc = keep.alloc(nv + 1)) || (rc = kpos.alloc(nv + 1))) return rc;
    const dim3 gv((unsigned)((nv + 255) / 256));
    hipLaunchKernelGGL(ror_cell_kernel, gv, dim3(256), 0, st, cen.p, nv, mn, 1.0 / radius, dims, cell.p);
    hipLaunchKernelGGL(iota_kernel, gv, dim3(256), 0, st, ids.p, nv);
    size_t tb = 0, tb2 = 0;
    int cell_bits = 1;
    while (cell_bits < 32 && ((size_t)1 << cell_bits) < ncell) cell_bits++;
    STOCS_HIP_CHECK(sort_pairs(NULL, tb, cell.p, cell_s.p, ids.p, ids_s.p, (size_t)nv, 0, (unsigned)cell_bits, st));
    STOCS_HIP_CHECK(exclusive_scan(NULL, tb2, keep.p, kpos.p, (size_t)nv + 1, st));
    if ((rc = tmp.alloc(std::max(tb, tb2)))) return rc;
    STOCS_HIP_CHECK(sort_pairs(tmp.p, tb, cell.p, cell_s.p, ids.p, ids_s.p, (size_t)nv, 0, (unsigned)cell_bits, st));
    hipLaunchKernelGGL(zero_u32x2_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, cstart.p, cend.p, ncell);
    hipLaunchKernelGGL(cell_start_kernel, gv, dim3(256), 0, st, cell_s.p, nv, cstart.p, cend.p);
    hipLaunchKernelGGL(ror_count_kernel, gv, dim3(256), 0, st, cen.p, nv, mn, 1.0 / radius, dims, radius, cstart.p, cend.p, ids_s.p, count.p, 10u);
    Buf<float4> on; Buf<float> op; Buf<int2> opx;
    if ((rc = on.alloc(nv)) || (rc = op.alloc(nv)) || (rc = opx.alloc(nv))) return rc;
    hipLaunchKernelGGL(scene_select_kernel, gv, dim3(256), 0, st, cen.p, count.p, nv, 10u, cam->fx, cam->cx, cam->fy, cam->cy, W, H, dC.p, class_threshold, dN.p,
                       keep.p, on.p, op.p, opx.p);
    // stable compaction on the device: destination = exclusive scan of the keep flags
    hipLaunchKernelGGL(zero_u32x2_kernel, dim3(1), dim3(256), 0, st, keep.p + nv, kpos.p + nv, (size_t)1);
    STOCS_HIP_CHECK(exclusive_scan(tmp.p, tb2, keep.p, kpos.p, (size_t)nv + 1, st));
    Buf<float> o_pos, o_nrm, o_prob; Buf<int32_t> o_px;
    if ((rc = o_pos.alloc((size_t)nv * 3)) || (rc = o_nrm.alloc((size_t)nv * 3)) || (rc = o_prob.alloc(nv)) || (rc = o_px.alloc((size_t)nv * 2))) return rc;
    hipLaunchKernelGGL(scene_pack_kernel, gv, dim3(256), 0, st, cen.p, on.p, op.p, opx.p, keep.p, kpos.p, nv, o_pos.p, o_nrm.p, o_prob.p, o_px.p);
    STOCS_HIP_CHECK(hipGetLastError());
    tick("outlier removal+select");
    uint32_t* m_pin = (uint32_t*)pin_small + 12;
    *m_pin = 0u;
    STOCS_HIP_CHECK(hipMemcpyAsync(m_pin, kpos.p + nv, 4, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t m = *m_pin;
    const size_t mc = std::min<size_t>(m, (size_t)std::max(cap, 0));
    // (the images in the staging block were consumed before the count came back: the block is free for the cloud)
    char* h_pos = pin; char* h_nrm = pin + 12 * mc; char* h_prob = pin + 24 * mc; char* h_px = pin + 28 * mc;
    if (mc) {
        if (pos3) STOCS_HIP_CHECK(hipMemcpyAsync(h_pos, o_pos.p, 12 * mc, hipMemcpyDeviceToHost, st));
        if (nrm3) STOCS_HIP_CHECK(hipMemcpyAsync(h_nrm, o_nrm.p, 12 * mc, hipMemcpyDeviceToHost, st));
        if (prob) STOCS_HIP_CHECK(hipMemcpyAsync(h_prob, o_prob.p, 4 * mc, hipMemcpyDeviceToHost, st));
        if (pixel2) STOCS_HIP_CHECK(hipMemcpyAsync(h_px, o_px.p, 8 * mc, hipMemcpyDeviceToHost, st));
    }
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    if (mc) {
        if (pos3) memcpy(pos3, h_pos, 12 * mc);
        if (nrm3) memcpy(nrm3, h_nrm, 12 * mc);
        if (prob) memcpy(prob, h_prob, 4 * mc);
        if (pixel2) memcpy(pixel2, h_px, 8 * mc);
    }
    tick("download");
    *n_out = (int)m;
    return (int)m > cap ? STOCS_ERR_CAPACITY : STOCS_OK;
}

int stocs_preprocess_model(const float* raw_pos3, int n_raw, float normal_radius, float voxel_size, float model_scale, int device,
                           float* pos3, float* nrm3, int cap, int* n_out) {
    if (!raw_pos3 || n_raw <= 0 || !n_out || !(normal_radius > 0) || !(voxel_size > 0)) return STOCS_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available: this library has no CPU fallback"); return STOCS_ERR_NO_DEVICE; }
    if (device >= 0) STOCS_HIP_CHECK(hipSetDevice(device));
    { int rc0 = workspace_begin(device); if (rc0) return rc0; }
    hipStream_t st = NULL;
    std::vector<float4> hp((size_t)n_raw);
    for (int i = 0; i < n_raw; ++i) hp[i] = make_float4(raw_pos3[3 * i], raw_pos3[3 * i + 1], raw_pos3[3 * i + 2], 0.f);
    Buf<float4> dP, dN;
    int rc;
    if ((rc = dP.alloc(n_raw)) || (rc = dN.alloc(n_raw))) return rc;
    STOCS_HIP_CHECK(hipMemcpy(dP.p, hp.data(), sizeof(float4) * (size_t)n_raw, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(model_normals_kernel, dim3((unsigned)((n_raw + 255) / 256)), dim3(256), 0, st, dP.p, n_raw, (double)normal_radius, dN.p);
    STOCS_HIP_CHECK(hipGetLastError());
    // drop points without a normal before the voxel grid (the numpy restatement does the same; PCL would
    // average NaNs into the leaf and rgbd::load_ply_model would then drop the whole leaf, rgbd.cpp:22)
    std::vector<float4> hn((size_t)n_raw);
    STOCS_HIP_CHECK(hipMemcpy(hn.data(), dN.p, sizeof(float4) * (size_t)n_raw, hipMemcpyDeviceToHost));
    std::vector<float4> kp, kn;
    for (int i = 0; i < n_raw; ++i)
        if (hn[i].x == hn[i].x && hn[i].y == hn[i].y && hn[i].z == hn[i].z) { kp.push_back(hp[i]); kn.push_back(hn[i]); }
    const int nk = (int)kp.size();
    *n_out = 0;
    if (nk == 0) return STOCS_OK;
    Buf<float4> dP2, dN2, cen, ext;
    if ((rc = dP2.alloc(nk)) || (rc = dN2.alloc(nk))) return rc;
    STOCS_HIP_CHECK(hipMemcpy(dP2.p, kp.data(), sizeof(float4) * (size_t)nk, hipMemcpyHostToDevice));
    STOCS_HIP_CHECK(hipMemcpy(dN2.p, kn.data(), sizeof(float4) * (size_t)nk, hipMemcpyHostToDevice));
    int nv = 0;
    if ((rc = voxel_grid_device(dP2.p, dN2.p, nk, (double)voxel_size, cen, ext, &nv, st))) return rc;   // stocs.cpp:54-57
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<float4> hc((size_t)std::max(nv, 1)), he((size_t)std::max(nv, 1));
    STOCS_HIP_CHECK(hipMemcpy(hc.data(), cen.p, sizeof(float4) * (size_t)nv, hipMemcpyDeviceToHost));
    STOCS_HIP_CHECK(hipMemcpy(he.data(), ext.p, sizeof(float4) * (size_t)nv, hipMemcpyDeviceToHost));
    int m = 0;
    for (int i = 0; i < nv; ++i) {
        const V3 nn = mk3(he[i].x, he[i].y, he[i].z);
        const double len = sqrt((double)nn.x * nn.x + (double)nn.y * nn.y + (double)nn.z * nn.z);
        if (!(len > 0) || !(len == len)) continue;   // load_ply_model keeps finite normals only (rgbd.cpp:22)
        if (m < cap) {
            if (pos3) { pos3[3 * m] = hc[i].x * model_scale; pos3[3 * m + 1] = hc[i].y * model_scale; pos3[3 * m + 2] = hc[i].z * model_scale; }
            if (nrm3) { nrm3[3 * m] = (float)(nn.x / len); nrm3[3 * m + 1] = (float)(nn.y / len); nrm3[3 * m + 2] = (float)(nn.z / len); }
        }
        m++;
    }
    *n_out = m;
    return m > cap ? STOCS_ERR_CAPACITY : STOCS_OK;
}

}  // extern "C"
