// sample.hip -- stochastic base sampling on the device (HOT LOOP A of the reference).
// Replaces
//   sample_point_from_distribution                        reference src/stocs.cpp:133-148
//   stocs_estimator::sample_class_base                    stocs.cpp:363-519
//   stocs_estimator::sample_instance_base (+prune_edge_pixels, generate_segmentation_mask)
//                                                         stocs.cpp:559-751, 521-535; src/rgbd.cpp:314-367
//   segment_distance_and_invariants / try_sampled_base    stocs.cpp:155-222 / 224-268
//
// Class mode: all attempts are independent (every attempt restarts from the class prior,
// stocs.cpp:373-381), so the whole batch of attempts runs as seven launches: draw, pass 1, draw,
// pass 2, draw, pass 3, draw.  A pass is one thread per (attempt, scene point): the PPF of
// (base point, point) -- three double-precision atan2, see stocs_math.h -- one bit test in the
// dilated existence bitmap of the model index (replaces std::map::find), the geometric tests, and
// a 4-byte weight write; reads are coalesced float4 loads of the scene SoA.  A draw is one
// 1024-thread workgroup per attempt: 2^32 fixed-point weights, block scan, binary choice of the
// owning chunk, i.e. an exact, order-independent replacement of std::discrete_distribution with a
// counter-based RNG (documented divergence Q6).
// Instance mode is sequential across attempts by construction (compounding prior decay and the
// previous segment, stocs.cpp:572-580,626); its image-space flood fill stays on the host as in the
// reference, the PPF passes and draws run on the device.
#include <math.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <limits>
#include <queue>

#include "stocs_ctx.h"

namespace stocs {

struct IndexView {
    const uint32_t* bits;
    int tr, rot, NA, nD;
};

__device__ __forceinline__ bool ppf_exists(const IndexView& ix, const int* K) {
    // lookup(K) is empty when K0 <= 5 or an angle key is negative (rgbd.cpp:136)
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return false;
    const int kd = K[0] / ix.tr, k1 = K[1] / ix.rot, k2 = K[2] / ix.rot, k3 = K[3] / ix.rot;
    if (kd >= ix.nD || k1 >= ix.NA || k2 >= ix.NA || k3 >= ix.NA) return false;
    const uint32_t key = ppf_pack(kd, k1, k2, k3, ix.NA);
    return (ix.bits[key >> 5] >> (key & 31)) & 1u;
}

struct PassArgs {
    const float4* spos;   // centred scene position (w unused here)
    const float4* snrm;   // unit normal
    int S;
    IndexView ix;
    float plane_threshold, min_distance_base;
    float ang_dot_hi, ang_dot_lo;
};

// w[b][i] = class_prob[i] for every attempt: "every base will start from the prior" (stocs.cpp:372-381)
__global__ __launch_bounds__(256) void init_weights_kernel(const float* __restrict__ cls, int S, float* __restrict__ w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) w[(size_t)blockIdx.y * S + i] = cls[i];
}

// PASS = 1: stocs.cpp:395-407   PASS = 2: stocs.cpp:424-442   PASS = 3: stocs.cpp:456-497
// whether pass PASS zeroes the weight of scene point i for the base points b1, b2, b3 chosen so far
template <int PASS>
__device__ __forceinline__ bool pass_zeroes(const PassArgs& a, int b1, int b2, int b3, int i) {
    const int cur = PASS == 1 ? b1 : (PASS == 2 ? b2 : b3);
    const float4 pc4 = a.spos[cur], nc4 = a.snrm[cur];
    const float4 pi4 = a.spos[i], ni4 = a.snrm[i];
    const V3 pc = mk3(pc4.x, pc4.y, pc4.z), nc = mk3(nc4.x, nc4.y, nc4.z);
    const V3 pi = mk3(pi4.x, pi4.y, pi4.z), ni = mk3(ni4.x, ni4.y, ni4.z);
    int K[4];
    ppf_compute(pc, nc, pi, ni, a.ix.tr, a.ix.rot, K);
    bool zero = !ppf_exists(a.ix, K) || i == cur;
    if (PASS == 2) {
        const float4 p14 = a.spos[b1];
        const V3 p1 = mk3(p14.x, p14.y, p14.z);
        const V3 v_1 = normalized3(pc - p1);
        const V3 v_2 = normalized3(pi - p1);
        const float d = dot3(v_1, v_2);
        // min(a, 180-a) < threshold with a = acos(d)*180/pi, as exact thresholds on d (NaN never rejects)
        zero = zero || (d >= a.ang_dot_hi && d <= 1.0f) || (d <= a.ang_dot_lo && d >= -1.0f);
    }
    if (PASS == 3) {
        const float4 p14 = a.spos[b1], p24 = a.spos[b2];
        const double x1 = p14.x, y1 = p14.y, z1 = p14.z;
        const double x2 = p24.x, y2 = p24.y, z2 = p24.z;
        const double x3 = pc4.x, y3 = pc4.y, z3 = pc4.z;
        const float denom = (float)(-x3 * y2 * z1 + x2 * y3 * z1 + x3 * y1 * z2 - x1 * y3 * z2 - x2 * y1 * z3 + x1 * y2 * z3);
        float planar_distance = 10000.0f;
        if (denom != 0) {
            const float A = (float)((-y2 * z1 + y3 * z1 + y1 * z2 - y3 * z2 - y1 * z3 + y2 * z3) / (double)denom);
            const float B = (float)((x2 * z1 - x3 * z1 - x1 * z2 + x3 * z2 + x1 * z3 - x2 * z3) / (double)denom);
            const float C = (float)((-x2 * y1 + x3 * y1 + x1 * y2 - x3 * y2 - x1 * y3 + x2 * y3) / (double)denom);
            const double v = (double)((A * pi.x + B * pi.y) + C * pi.z) - 1.0;
            planar_distance = (float)(v < 0 ? -v : v);
        }
        const V3 p1 = mk3(p14.x, p14.y, p14.z), p2 = mk3(p24.x, p24.y, p24.z);
        zero = zero || planar_distance > a.plane_threshold || norm3(pi - p1) < a.min_distance_base ||
               norm3(pi - p2) < a.min_distance_base || norm3(pi - pc) < a.min_distance_base;
    }
    return zero;
}

template <int PASS>
__global__ __launch_bounds__(256) void pass_kernel(PassArgs a, const int32_t* __restrict__ bidx, const int32_t* __restrict__ fail,
                                                   float* __restrict__ w) {
    const int b = blockIdx.y;
    if (fail[b]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.S) return;
    if (pass_zeroes<PASS>(a, bidx[b * 4 + 0], bidx[b * 4 + 1], bidx[b * 4 + 2], i)) w[(size_t)b * a.S + i] = 0.0f;
}

// seeded weighted draw by one workgroup of 1024 threads: index into wb, or -1 when every weight is zero
// ("FAILED SAMPLING:: Zero probability returned", stocs.cpp:386-389).  Every thread returns the same value.
__device__ __forceinline__ int draw_block(const float* __restrict__ wb, int S, uint64_t r64, uint64_t* sh /*1024*/, uint64_t* sh_total, int* sh_pick) {
    const int t = threadIdx.x;
    const int chunk = (S + 1023) / 1024;
    const int lo = min(S, t * chunk), hi = min(S, lo + chunk);
    uint64_t local = 0;
    for (int i = lo; i < hi; ++i) local += weight_fix(wb[i]);
    sh[t] = local;
    __syncthreads();
    // Hillis-Steele inclusive scan (integer adds: exact, order-independent)
    for (int off = 1; off < 1024; off <<= 1) {
        uint64_t v = (t >= off) ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    if (t == 1023) { *sh_total = sh[t]; *sh_pick = -1; }
    __syncthreads();
    const uint64_t total = *sh_total;
    if (total != 0) {
        const uint64_t r = mulhi64(r64, total);
        const uint64_t incl = sh[t], excl = incl - local;
        if (r >= excl && r < incl) {  // exactly one thread (local > 0)
            uint64_t c = excl;
            int pick = -1;
            for (int i = lo; i < hi; ++i) {
                c += weight_fix(wb[i]);
                if (c > r) { pick = i; break; }
            }
            *sh_pick = pick;
        }
    }
    __syncthreads();
    const int pick = *sh_pick;
    __syncthreads();   // sh / sh_pick may be reused by the next draw
    return pick;
}

// one workgroup per attempt
__global__ __launch_bounds__(1024) void draw_kernel(const float* __restrict__ w, size_t stride, int S, uint64_t seed,
                                                    uint64_t first_attempt, uint64_t k, const uint64_t* __restrict__ r_explicit,
                                                    int slot, int32_t* __restrict__ bidx, int32_t* __restrict__ fail) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t sh_total;
    __shared__ int sh_pick;
    const int b = blockIdx.x;
    if (fail[b]) return;
    const uint64_t r64 = r_explicit ? r_explicit[b] : rng64(seed, first_attempt + (uint64_t)b, k);
    const int pick = draw_block(w + (size_t)b * stride, S, r64, sh, &sh_total, &sh_pick);
    if (threadIdx.x == 0) {
        bidx[b * 4 + slot] = pick;
        if (pick < 0) fail[b] = 1;
    }
}

// ---- host helpers ------------------------------------------------------------------------------

// stocs.cpp:155-222 with VectorType = float vector, Scalar = double (deduced at the call :237-244)
__host__ __device__ static double seg_dist_inv(V3 p1, V3 p2, V3 q1, V3 q2, double& invariant1, double& invariant2) {
    const double kSmallNumber = 0.0001;
    const V3 u = p2 - p1, v = q2 - q1, w = p1 - q1;
    const double a = dot3(u, u), b = dot3(u, v), c = dot3(v, v), d = dot3(u, w), e = dot3(v, w);
    const double f = a * c - b * b;
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
    const float i1 = (float)invariant1, i2 = (float)invariant2;  // double * Vector3f narrows the scalar
    return (double)norm3((w + (i1 * u)) - (i2 * v));
}

// stocs.cpp:224-268
__host__ __device__ static bool try_sampled_base(const V3 base[4], float& invariant1, float& invariant2, int ids[4]) {
    float min_distance = 3.402823466e+38f;   // std::numeric_limits<float>::max()
    int best1 = -1, best2 = -1, best3 = -1, best4 = -1;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            if (i == j) continue;
            int k = 0;
            while (k == i || k == j) k++;
            int l = 0;
            while (l == i || l == j || l == k) l++;
            double li1, li2;
            const float segment_distance = (float)seg_dist_inv(base[i], base[j], base[k], base[l], li1, li2);
            if (segment_distance < min_distance) {  // strict <: first minimum wins (Q19)
                min_distance = segment_distance;
                best1 = i; best2 = j; best3 = k; best4 = l;
                invariant1 = (float)li1;
                invariant2 = (float)li2;
            }
        }
    if (best1 < 0 || best2 < 0 || best3 < 0 || best4 < 0) return false;
    const int tmp[4] = {ids[0], ids[1], ids[2], ids[3]};
    ids[0] = tmp[best1]; ids[1] = tmp[best2]; ids[2] = tmp[best3]; ids[3] = tmp[best4];
    return true;
}

// rows 6-7 on the device: ordered base + invariants of one attempt (try_sampled_base, stocs.cpp:224-268)
struct BaseOut { int32_t ids[4]; float inv[2]; int32_t valid; int32_t pad; };

__device__ __forceinline__ void finalize_one(const float4* __restrict__ spos, const int32_t* bidx4, int fail, BaseOut* o) {
    int ids[4] = {bidx4[0], bidx4[1], bidx4[2], bidx4[3]};
    float i1 = 0, i2 = 0;
    bool ok = !fail && ids[0] >= 0 && ids[1] >= 0 && ids[2] >= 0 && ids[3] >= 0;
    if (ok) {
        V3 base[4];
        for (int k = 0; k < 4; ++k) { const float4 p = spos[ids[k]]; base[k] = mk3(p.x, p.y, p.z); }
        ok = try_sampled_base(base, i1, i2, ids);
    }
    for (int k = 0; k < 4; ++k) o->ids[k] = ids[k];
    o->inv[0] = i1; o->inv[1] = i2;
    o->valid = ok ? 1 : 0;
    o->pad = 0;
}

__global__ __launch_bounds__(64) void finalize_bases_kernel(const float4* __restrict__ spos, const int32_t* __restrict__ bidx,
                                                            const int32_t* __restrict__ fail, int nB, BaseOut* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nB) finalize_one(spos, bidx + 4 * b, fail[b], out + b);
}

struct SampleBuffers {
    BaseOut* res;    // nB
    float* w;        // nB * S
    float* cls;      // S (class probabilities)
    int32_t* bidx;   // nB * 4
    int32_t* fail;   // nB
    uint64_t* rexp;  // nB
};

static int carve(stocs_ctx* c, int nB, SampleBuffers* sb) {
    const size_t S = (size_t)std::max(c->nS, 1);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t bw = al((size_t)nB * S * 4), bc = al(S * 4), bi = al((size_t)nB * 16), bf = al((size_t)nB * 4), br = al((size_t)nB * 8),
                 bo = al((size_t)nB * sizeof(BaseOut));
    int rc = ensure_scratch(c, bw + bc + bi + bf + br + bo);
    if (rc) return rc;
    char* p = (char*)c->d_scratch;
    sb->w = (float*)p; p += bw;          // weights, then bidx, then fail: one copy moves all three when nB == 1
    sb->bidx = (int32_t*)p; p += bi;
    sb->fail = (int32_t*)p; p += bf;
    sb->cls = (float*)p; p += bc;
    sb->rexp = (uint64_t*)p; p += br;
    sb->res = (BaseOut*)p;
    return STOCS_OK;
}

static PassArgs pass_args(const stocs_ctx* c) {
    PassArgs a;
    a.spos = c->d_spos; a.snrm = c->d_snrmw; a.S = c->nS;
    a.ix.bits = c->index.d_exists; a.ix.tr = c->index.tr; a.ix.rot = c->index.rot; a.ix.NA = c->index.NA; a.ix.nD = c->index.nD;
    a.plane_threshold = c->prm.plane_threshold; a.min_distance_base = c->prm.min_distance_base;
    a.ang_dot_hi = c->thr.ang_dot_hi; a.ang_dot_lo = c->thr.ang_dot_lo;
    return a;
}

static void launch_pass(stocs_ctx* c, int pass, int nB, const SampleBuffers& sb) {
    const PassArgs a = pass_args(c);
    const dim3 grid((unsigned)((c->nS + 255) / 256), (unsigned)nB);
    if (pass == 1) hipLaunchKernelGGL(pass_kernel<1>, grid, dim3(256), 0, c->stream, a, sb.bidx, sb.fail, sb.w);
    else if (pass == 2) hipLaunchKernelGGL(pass_kernel<2>, grid, dim3(256), 0, c->stream, a, sb.bidx, sb.fail, sb.w);
    else hipLaunchKernelGGL(pass_kernel<3>, grid, dim3(256), 0, c->stream, a, sb.bidx, sb.fail, sb.w);
}

static void launch_draw(stocs_ctx* c, int nB, const SampleBuffers& sb, uint64_t seed, uint64_t first_attempt, int k) {
    hipLaunchKernelGGL(draw_kernel, dim3((unsigned)nB), dim3(1024), 0, c->stream, sb.w, (size_t)c->nS, c->nS, seed, first_attempt,
                       (uint64_t)k, (const uint64_t*)NULL, k, sb.bidx, sb.fail);
}

// host bookkeeping of the attempts the device has finalised (finalize_one): outputs + the context's base set
static int record_bases(stocs_ctx* c, int nB, const BaseOut* res, int32_t* ids_out, float* inv_out, int32_t* valid_out) {
    for (int b = 0; b < nB; ++b) {
        if (ids_out) for (int k = 0; k < 4; ++k) ids_out[b * 4 + k] = res[b].ids[k];
        if (inv_out) { inv_out[b * 2] = res[b].inv[0]; inv_out[b * 2 + 1] = res[b].inv[1]; }
        if (valid_out) valid_out[b] = res[b].valid;
        if (res[b].valid) {
            BaseRec r;
            for (int k = 0; k < 4; ++k) r.ids[k] = res[b].ids[k];
            r.inv1 = res[b].inv[0]; r.inv2 = res[b].inv[1];
            c->bases.push_back(r);
        }
    }
    c->quad_off.clear();
    return STOCS_OK;
}

static int upload_class_prob(stocs_ctx* c, const SampleBuffers& sb) {
    STOCS_HIP_CHECK(hipMemcpyAsync(sb.cls, c->h_sprob.data(), (size_t)c->nS * 4, hipMemcpyHostToDevice, c->stream));
    return STOCS_OK;
}

static int sample_class(stocs_ctx* c, uint64_t seed, int first_attempt, int nB, int32_t* ids, float* inv, int32_t* valid) {
    SampleBuffers sb;
    int rc = carve(c, nB, &sb);
    if (rc) return rc;
    if ((rc = upload_class_prob(c, sb))) return rc;
    STOCS_HIP_CHECK(hipMemsetAsync(sb.fail, 0, (size_t)nB * 4, c->stream));
    STOCS_HIP_CHECK(hipMemsetAsync(sb.bidx, 0xFF, (size_t)nB * 16, c->stream));
    hipLaunchKernelGGL(init_weights_kernel, dim3((unsigned)((c->nS + 255) / 256), (unsigned)nB), dim3(256), 0, c->stream, sb.cls, c->nS, sb.w);
    for (int k = 0; k < 4; ++k) {
        launch_draw(c, nB, sb, seed, (uint64_t)first_attempt, k);
        if (k < 3) launch_pass(c, k + 1, nB, sb);
    }
    STOCS_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(finalize_bases_kernel, dim3((unsigned)((nB + 63) / 64)), dim3(64), 0, c->stream, c->d_spos, sb.bidx, sb.fail, nB, sb.res);
    STOCS_HIP_CHECK(hipGetLastError());
    std::vector<BaseOut> res((size_t)nB);
    STOCS_HIP_CHECK(hipMemcpyAsync(res.data(), sb.res, (size_t)nB * sizeof(BaseOut), hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return record_bases(c, nB, res.data(), ids, inv, valid);
}

// rgbd.cpp:314-367 with the PNG round trip through dbg/seg_mask_<n>.png replaced by seg_masks (Q14)
static std::shared_ptr<const std::vector<uint8_t> > generate_segmentation_mask(stocs_ctx* c, int prow, int pcol, float max_distance, int base_num) {
    const int W = c->prm.image_width, H = c->prm.image_height;
    const int segment_index = c->segmentation_buffer[(size_t)prow * W + pcol];
    if (segment_index != 0) return c->seg_masks[segment_index];   // the mask file of that attempt is read back (rgbd.cpp:322-326)
    std::shared_ptr<std::vector<uint8_t> > closed(new std::vector<uint8_t>((size_t)W * H, 0));
    std::vector<uint8_t>& closed_list = *closed;
    std::queue<std::pair<int, int> > open_list;
    open_list.push(std::make_pair(prow, pcol));
    while (!open_list.empty()) {
        const std::pair<int, int> curr = open_list.front();
        closed_list[(size_t)curr.first * W + curr.second] = 255;
        c->segmentation_buffer[(size_t)curr.first * W + curr.second] = (uint8_t)base_num;
        open_list.pop();
        for (int i = curr.first - 1; i <= curr.first + 1; ++i)
            for (int j = curr.second - 1; j <= curr.second + 1; ++j) {
                if (i < 0 || j < 0 || i >= H || j >= W) continue;
                const float edge_probability = (float)(255.0 - c->edge_map[(size_t)i * W + j]) / 255.0;
                const int expanded = (int)closed_list[(size_t)i * W + j];
                const float dist = (float)sqrt(pow((double)(prow - i), 2) + pow((double)(pcol - j), 2));
                if (expanded == 0 && edge_probability == 0 && dist < max_distance) {
                    open_list.push(std::make_pair(i, j));
                    closed_list[(size_t)i * W + j] = 255;
                    c->segmentation_buffer[(size_t)i * W + j] = (uint8_t)base_num;
                }
            }
    }
    return closed;
}

static int refresh_class_prob_on_device(stocs_ctx* c) {
    // the LCP adds class_probability_, which instance-mode sampling decays in place (Q8)
    std::vector<float4> a(c->nS), b(c->nS);
    for (int i = 0; i < c->nS; ++i) {
        a[i] = make_float4(c->h_spos[i].x, c->h_spos[i].y, c->h_spos[i].z, c->h_sprob[i]);
        b[i] = make_float4(c->h_snrm[i].x, c->h_snrm[i].y, c->h_snrm[i].z, c->h_sprob[i]);
    }
    STOCS_HIP_CHECK(hipMemcpyAsync(c->d_spos, a.data(), (size_t)c->nS * 16, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(c->d_snrmw, b.data(), (size_t)c->nS * 16, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

static thread_local double g_t_inst[6];   // STOCS_DEBUG_TIMING accumulators
static inline double now_s_() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define TSEC(k) { const double t_ = now_s_(); g_t_inst[k] += t_ - tprev_; tprev_ = t_; }
static int sample_instance_one(stocs_ctx* c, uint64_t seed, int attempt, float dispersion, int base_num, int32_t* ids, float* inv, int32_t* valid) {
    const int S = c->nS, W = c->prm.image_width;
    double tprev_ = now_s_();
    SampleBuffers sb;
    int rc = carve(c, 1, &sb);
    if (rc) return rc;
    std::vector<float> w(S);
    for (int i = 0; i < S; ++i) {  // stocs.cpp:572-580 (compounding decay) + prune_edge_pixels :521-535
        const size_t px = (size_t)c->h_spix[2 * i] * W + c->h_spix[2 * i + 1];
        if (c->previous_segment && (*c->previous_segment)[px]) c->h_sprob[i] = dispersion * c->h_sprob[i];
        w[i] = c->h_sprob[i];
        const float edge_probability = (float)(255.0 - c->edge_map[px]) / 255.0;
        if (edge_probability == 1) w[i] = 0;
    }
    TSEC(0)
    // round trip 1: weights up, draw point 1, pass 1, weights + (bidx, fail) back in one copy
    const size_t span = (size_t)((char*)sb.fail - (char*)sb.w) + 4;   // w .. fail, contiguous (carve)
    // pinned staging (grown on demand, owned by the context): the small copies of every attempt go straight over DMA
    if (c->pin_bytes < span + 64) {
        if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = NULL; c->pin_bytes = 0; }
        STOCS_HIP_CHECK(hipHostMalloc(&c->h_pin, span + 4096, hipHostMallocDefault));
        c->pin_bytes = span + 4096;
    }
    struct { char* p; char* data() { return p; } } stage = {(char*)c->h_pin};
    {
        // weights + (bidx = -1, fail = 0) go up in ONE copy (they are contiguous, carve); draw and pass are separate
        // launches so that the double-precision pass runs on the whole chip (a fused one-workgroup kernel was 25 % slower)
        memcpy(stage.data(), w.data(), (size_t)S * 4);
        const int32_t init_bidx[4] = {-1, -1, -1, -1}, init_fail = 0;
        memcpy(stage.data() + ((char*)sb.bidx - (char*)sb.w), init_bidx, 16);
        memcpy(stage.data() + ((char*)sb.fail - (char*)sb.w), &init_fail, 4);
        STOCS_HIP_CHECK(hipMemcpyAsync(sb.w, stage.data(), span, hipMemcpyHostToDevice, c->stream));
        launch_draw(c, 1, sb, seed, (uint64_t)attempt, 0);
        launch_pass(c, 1, 1, sb);
    }
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(hipMemcpyAsync(stage.data(), sb.w, span, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    int32_t bidx[4], fail = 0;
    memcpy(w.data(), stage.data(), (size_t)S * 4);
    memcpy(bidx, stage.data() + ((char*)sb.bidx - (char*)sb.w), 16);
    memcpy(&fail, stage.data() + ((char*)sb.fail - (char*)sb.w), 4);
    TSEC(1)
    BaseOut res;
    memset(&res, 0, sizeof(res));
    res.ids[0] = res.ids[1] = res.ids[2] = res.ids[3] = -1;   // "FAILED SAMPLING": no base, nothing to compute
    if (fail || bidx[0] < 0) return record_bases(c, 1, &res, ids, inv, valid);
    const int b1 = bidx[0];
    float max_pixel_distance = 0;  // stocs.cpp:610-618
    for (int i = 0; i < S; ++i)
        if (w[i] != 0) {
            const float dist = (float)sqrt(pow((double)(c->h_spix[2 * b1] - c->h_spix[2 * i]), 2) + pow((double)(c->h_spix[2 * b1 + 1] - c->h_spix[2 * i + 1]), 2));
            if (dist > max_pixel_distance) max_pixel_distance = dist;
        }
    TSEC(2)
    const std::shared_ptr<const std::vector<uint8_t> > mask_p = generate_segmentation_mask(c, c->h_spix[2 * b1], c->h_spix[2 * b1 + 1], max_pixel_distance, base_num);
    const std::vector<uint8_t>& mask = *mask_p;
    TSEC(3)
    if ((int)c->seg_masks.size() <= base_num) c->seg_masks.resize(base_num + 1);
    c->seg_masks[base_num] = mask_p;    // cv::imwrite(seg_mask_<n>.png), stocs.cpp:625
    c->previous_segment = mask_p;       // stocs.cpp:626
    c->last_segment.clear();
    for (int i = 0; i < S; ++i)         // stocs.cpp:628-638: survivors inside the mask form `segment`, the others are zeroed
        if (w[i] != 0) {
            if (mask[(size_t)c->h_spix[2 * i] * W + c->h_spix[2 * i + 1]]) c->last_segment.push_back(i);
            else w[i] = 0;
        }
    TSEC(4)
    // round trip 2: filtered weights up, draw 2, pass 2, draw 3, pass 3, draw 4, base finalised on the device, result back
    memcpy(stage.data(), w.data(), (size_t)S * 4);
    STOCS_HIP_CHECK(hipMemcpyAsync(sb.w, stage.data(), (size_t)S * 4, hipMemcpyHostToDevice, c->stream));
    for (int k = 1; k < 4; ++k) {
        launch_draw(c, 1, sb, seed, (uint64_t)attempt, k);
        if (k < 3) launch_pass(c, k + 1, 1, sb);
    }
    hipLaunchKernelGGL(finalize_bases_kernel, dim3(1), dim3(64), 0, c->stream, c->d_spos, sb.bidx, sb.fail, 1, sb.res);
    STOCS_HIP_CHECK(hipGetLastError());
    BaseOut* pres = (BaseOut*)(stage.data() + (((size_t)S * 4 + 63) & ~(size_t)63));
    STOCS_HIP_CHECK(hipMemcpyAsync(pres, sb.res, sizeof(res), hipMemcpyDeviceToHost, c->stream));   // ordered base + invariants, finalised on the device
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    res = *pres;
    TSEC(5)
    return record_bases(c, 1, &res, ids, inv, valid);
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_sample_bases(stocs_ctx* c, int mode, uint64_t seed, int first_attempt, int n_attempts, float dispersion,
                       int32_t* base_ids4, float* inv2, int32_t* valid) {
    if (!c || n_attempts < 0 || first_attempt < 0 || (mode != 0 && mode != 1)) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->index.built) { set_error("stocs_sample_bases: PPF index not built"); return STOCS_ERR_STATE; }
    if (n_attempts == 0) return STOCS_OK;
    if (c->nS == 0) {  // the reference would index an empty vector here (stocs.cpp:386); report no bases
        for (int b = 0; b < n_attempts; ++b) if (valid) valid[b] = 0;
        return STOCS_OK;
    }
    if (mode == 0) return sample_class(c, seed, first_attempt, n_attempts, base_ids4, inv2, valid);
    if (n_attempts + first_attempt > 254) { set_error("instance mode labels segments with a u8 (<= 254 attempts, Q14)"); return STOCS_ERR_INVALID; }
    for (int b = 0; b < n_attempts; ++b) {
        int rc = sample_instance_one(c, seed, first_attempt + b, dispersion, first_attempt + b + 1, base_ids4 ? base_ids4 + 4 * b : NULL,
                                     inv2 ? inv2 + 2 * b : NULL, valid ? valid + b : NULL);
        if (rc) return rc;
    }
    if (getenv("STOCS_DEBUG_TIMING"))
        fprintf(stderr, "[stocs instance] cumulative ms: weights %.2f | draw0+pass1+sync %.2f | maxdist %.2f | flood fill %.2f | mask copies+filter %.2f | draws 1-3 + sync %.2f\n",
                g_t_inst[0] * 1e3, g_t_inst[1] * 1e3, g_t_inst[2] * 1e3, g_t_inst[3] * 1e3, g_t_inst[4] * 1e3, g_t_inst[5] * 1e3);
    return refresh_class_prob_on_device(c);
}

// `segment` of the last instance-mode attempt (stocs.cpp:628-638): the scene points that survived pass 1 inside the
// segmentation mask, in scene order
int stocs_get_segment(const stocs_ctx* c, int32_t* scene_idx, int cap, int* n) {
    if (!c || !n) return STOCS_ERR_INVALID;
    *n = (int)c->last_segment.size();
    if (!scene_idx) return STOCS_OK;
    for (int i = 0; i < *n && i < cap; ++i) scene_idx[i] = c->last_segment[i];
    return (*n > cap) ? STOCS_ERR_CAPACITY : STOCS_OK;
}

int stocs_reset_trial(stocs_ctx* c) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->h_sprob = c->h_sprob0;
    c->previous_segment.reset();
    std::fill(c->segmentation_buffer.begin(), c->segmentation_buffer.end(), 0);
    c->seg_masks.clear();
    c->last_segment.clear();
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1;
    return refresh_class_prob_on_device(c);
}

int stocs_set_bases(stocs_ctx* c, int n, const int32_t* ids, const float* inv) {
    if (!c || n < 0 || (n && (!ids || !inv))) return STOCS_ERR_INVALID;
    for (int i = 0; i < 4 * n; ++i)
        if (ids[i] < 0 || ids[i] >= c->nS) { set_error("stocs_set_bases: scene index out of range"); return STOCS_ERR_INVALID; }
    c->bases.clear();
    c->quad_off.clear();
    for (int i = 0; i < n; ++i) {
        BaseRec b;
        for (int k = 0; k < 4; ++k) b.ids[k] = ids[4 * i + k];
        b.inv1 = inv[2 * i]; b.inv2 = inv[2 * i + 1];
        c->bases.push_back(b);
    }
    return STOCS_OK;
}
int stocs_clear_bases(stocs_ctx* c) {
    if (!c) return STOCS_ERR_INVALID;
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1;
    return STOCS_OK;
}
int stocs_num_bases(const stocs_ctx* c) { return c ? (int)c->bases.size() : STOCS_ERR_INVALID; }

int stocs_class_pass(stocs_ctx* c, int pass, const int32_t* b3, const float* w_in, float* w_out) {
    if (!c || pass < 1 || pass > 3 || !b3 || !w_in || !w_out) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->index.built) { set_error("stocs_class_pass: PPF index not built"); return STOCS_ERR_STATE; }
    for (int k = 0; k < pass; ++k) if (b3[k] < 0 || b3[k] >= c->nS) return STOCS_ERR_INVALID;
    SampleBuffers sb;
    int rc = carve(c, 1, &sb);
    if (rc) return rc;
    int32_t bidx[4] = {b3[0], pass >= 2 ? b3[1] : 0, pass >= 3 ? b3[2] : 0, -1};
    STOCS_HIP_CHECK(hipMemcpyAsync(sb.w, w_in, (size_t)c->nS * 4, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(sb.bidx, bidx, 16, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemsetAsync(sb.fail, 0, 4, c->stream));
    launch_pass(c, pass, 1, sb);
    STOCS_HIP_CHECK(hipGetLastError());
    STOCS_HIP_CHECK(hipMemcpyAsync(w_out, sb.w, (size_t)c->nS * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

int stocs_try_sampled_base(stocs_ctx* c, int32_t* ids4, float* inv2, int* valid) {
    if (!c || !ids4 || !inv2 || !valid) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    for (int k = 0; k < 4; ++k) if (ids4[k] < 0 || ids4[k] >= c->nS) return STOCS_ERR_INVALID;
    SampleBuffers sb;
    int rc = carve(c, 1, &sb);
    if (rc) return rc;
    STOCS_HIP_CHECK(hipMemcpyAsync(sb.bidx, ids4, 16, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemsetAsync(sb.fail, 0, 4, c->stream));
    hipLaunchKernelGGL(finalize_bases_kernel, dim3(1), dim3(64), 0, c->stream, c->d_spos, sb.bidx, sb.fail, 1, sb.res);
    STOCS_HIP_CHECK(hipGetLastError());
    BaseOut res;
    STOCS_HIP_CHECK(hipMemcpyAsync(&res, sb.res, sizeof(res), hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *valid = res.valid;
    for (int k = 0; k < 4; ++k) ids4[k] = res.ids[k];
    inv2[0] = res.inv[0]; inv2[1] = res.inv[1];
    return STOCS_OK;
}

int stocs_draw(stocs_ctx* c, const float* w, int n, uint64_t r64, int* index) {
    if (!c || !w || n <= 0 || !index) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    int rc = ensure_scratch(c, al((size_t)n * 4) + 1024);
    if (rc) return rc;
    char* p = (char*)c->d_scratch;
    float* dw = (float*)p; p += al((size_t)n * 4);
    int32_t* bidx = (int32_t*)p; p += 256;
    int32_t* fail = (int32_t*)p; p += 256;
    uint64_t* rexp = (uint64_t*)p;
    STOCS_HIP_CHECK(hipMemcpyAsync(dw, w, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(rexp, &r64, 8, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemsetAsync(fail, 0, 4, c->stream));
    STOCS_HIP_CHECK(hipMemsetAsync(bidx, 0xFF, 16, c->stream));
    hipLaunchKernelGGL(draw_kernel, dim3(1), dim3(1024), 0, c->stream, dw, (size_t)n, n, (uint64_t)0, (uint64_t)0, (uint64_t)0, rexp, 0, bidx, fail);
    STOCS_HIP_CHECK(hipGetLastError());
    int32_t out[4];
    STOCS_HIP_CHECK(hipMemcpyAsync(out, bidx, 16, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *index = out[0];
    return STOCS_OK;
}

}  // extern "C"
