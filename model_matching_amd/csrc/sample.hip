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
// pass 2, draw, pass 3, draw.  A pass is one thread per (attempt, scene point): the PPF key of
// (base point, point) -- a float filter in front of the reference's double atan2 arithmetic, see
// ppf_key_fast below -- one bit test in the dilated existence bitmap of the model index (replaces
// std::map::find), the geometric tests, and a 4-byte weight write; reads are coalesced float4 loads
// of the scene SoA.  A draw is one 1024-thread workgroup per attempt: 2^32 fixed-point weights,
// wavefront scans, i.e. an exact, order-independent replacement of std::discrete_distribution with
// a counter-based RNG (documented divergence Q6).
// Instance mode is sequential across attempts by construction (compounding prior decay and the
// previous segment, stocs.cpp:572-580,626): one kernel of two pipelined workgroups runs all its attempts
// on the device, image-space flood fill included (instance_attempts_kernel below); the host only reads the results.
#include <math.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <limits>

#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

struct IndexView {
    const uint32_t* bits;
    int tr, rot, NA, nD;
    // a pair can only have a key when its squared length lies in [sq_lo, sq_hi): see set_distance_thresholds
    float sq_lo, sq_hi;
    int far_certain;   // 0: beyond sq_hi the exact path decides (an index reaching past a kilometre)
};

__device__ __forceinline__ bool ppf_exists(const IndexView& ix, const int* K) {
    // lookup(K) is empty when K0 <= 5 or an angle key is negative (rgbd.cpp:136)
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return false;
    const int kd = K[0] / ix.tr, k1 = K[1] / ix.rot, k2 = K[2] / ix.rot, k3 = K[3] / ix.rot;
    if (kd >= ix.nD || k1 >= ix.NA || k2 >= ix.NA || k3 >= ix.NA) return false;
    const uint32_t key = ppf_pack(kd, k1, k2, k3, ix.NA);
    return (ix.bits[key >> 5] >> (key & 31)) & 1u;
}
// packed key of the existence bitmap for the PPF K, or PPF_NO_KEY when lookup(K) is empty by construction (rgbd.cpp:136)
#define PPF_NO_KEY 0xFFFFFFFFu
__device__ __forceinline__ uint32_t ppf_pack_or_none(const IndexView& ix, const int* K) {
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return PPF_NO_KEY;
    const int kd = K[0] / ix.tr, k1 = K[1] / ix.rot, k2 = K[2] / ix.rot, k3 = K[3] / ix.rot;
    if (kd >= ix.nD || k1 >= ix.NA || k2 >= ix.NA || k3 >= ix.NA) return PPF_NO_KEY;
    return ppf_pack(kd, k1, k2, k3, ix.NA);
}
__device__ __forceinline__ bool ppf_key_present(const IndexView& ix, uint32_t key) {
    return key != PPF_NO_KEY && ((ix.bits[key >> 5] >> (key & 31)) & 1u);
}

// PPF key of (p1, p2) as ppf_compute gives it (reference rgbd.cpp:99-121), by a float filter with an exact fallback.
// The three angles of the reference are v = int(atan2(double, double) * 180 / pi), quantised by ppf_closest_bin
// (rgbd.cpp:85-97) to rot * k with k = floor((v - b) / rot) + 1, b = ceil(rot / 2): since rot * k + b is an integer,
// k = floor((a - b) / rot) + 1 for the untruncated angle a as well, and the key only changes where a crosses such a
// boundary.  The filter evaluates a in float with its own atan2 (v_rcp_f32 + an odd polynomial of degree 15, |error| <
// 2e-7 rad; with the roundings of the operands and of the conversion the float angle is within 1e-4 degree of the double
// one) and no integer division; when the float angle is more than PPF_MARGIN_DEG = 1e-3 degree away from every boundary
// the float key IS the reference's key; otherwise (about 0.1 % of the pairs, and anything non-finite) the double evaluation
// of stocs_math.h decides.  Distance bin: the same float arithmetic in both.  stocs_ppf_filter_check counts disagreements.
#define PPF_MARGIN_DEG 1e-3f
// atan2(y, x) in degrees for y >= 0 (a norm): [0, 180]; NaN when both are zero or infinite
__device__ __forceinline__ float atan2_deg_fast(float y, float x) {
    const float ax = fabsf(x);
    const float mx = fmaxf(ax, y), mn = fminf(ax, y);
    const float r = mn * __builtin_amdgcn_rcpf(mx);              // [0, 1]
    const float s = r * r;
    float p = -0.00458501698449254f;                             // atan(r) = r * P(r^2) on [0, 1], max error 1.8e-7 rad in float
    p = fmaf(p, s, 0.023815227672457695f);
    p = fmaf(p, s, -0.05878564342856407f);
    p = fmaf(p, s, 0.09857634454965591f);
    p = fmaf(p, s, -0.13995274901390076f);
    p = fmaf(p, s, 0.19964531064033508f);
    p = fmaf(p, s, -0.3333152234554291f);
    p = fmaf(p, s, 0.9999998211860657f);
    p *= r;
    if (y > ax) p = 1.5707963267948966f - p;
    if (x < 0.0f) p = 3.14159265358979f - p;
    return p * 57.29577951308232f;
}
// bin index k of one angle; true when the angle is clear of the bin boundaries
__device__ __forceinline__ bool ppf_angle_bin_fast(V3 cr, float x, float b, float inv_rot, float rot_f, int* k) {
    const float y = __builtin_amdgcn_sqrtf(sqn3(cr));           // 1 ulp is plenty in front of the margin
    const float a = atan2_deg_fast(y, x);
    const float q = (a - b) * inv_rot;
    const float fl = floorf(q), fq = q - fl;
    *k = (int)fl + 1;
    return fminf(fq, 1.0f - fq) * rot_f > PPF_MARGIN_DEG && x == x && y == y;   // false for NaN
}
// distance bin index of int(norm * 1000) (rgbd.cpp:103) for f = norm * 1000 < 2^20, ppf_closest_bin without the division:
// k = floor((v - ceil(tr / 2)) / tr) + 1 = (v - ceil(tr / 2) + tr) / tr
__device__ __forceinline__ int ppf_distance_bin(int tr, float f) {
    const int n = (int)f - (tr + 1) / 2 + tr;
    int kd = (int)((float)n * (1.0f / (float)tr));               // within one of the quotient for n < 2^21: one correction step
    const int rem = n - kd * tr;
    if (rem < 0) --kd; else if (rem >= tr) ++kd;
    return kd;
}
// false when the distance component alone leaves no key (K0 <= 5, rgbd.cpp:136, or beyond the index), decided on the
// squared length (set_distance_thresholds); NaN, and lengths beyond a kilometre, are left to the exact path: true
__device__ __forceinline__ bool ppf_distance_may_have_key(const IndexView& ix, V3 u) {
    const float sq = sqn3(u);
    return (sq >= ix.sq_lo && sq < ix.sq_hi) || !(sq == sq) || (!ix.far_certain && sq >= ix.sq_hi);
}
// packed key from the float evaluation; *certain when it IS the key of the reference's arithmetic
__device__ __forceinline__ uint32_t ppf_key_fast(const IndexView& ix, V3 p1, V3 n1, V3 p2, V3 n2, bool* certain) {
    const float rot_f = (float)ix.rot, inv_rot = 1.0f / rot_f, b = (float)((ix.rot + 1) / 2);   // ceil(rot / 2): remainders >= b round up
    const V3 u = p1 - p2;
    const float sq = sqn3(u);
    if (!(sq >= ix.sq_lo && sq < ix.sq_hi)) {                    // no key whatever the angles -- or NaN / beyond a kilometre: the exact path
        *certain = sq == sq && (ix.far_certain || sq < ix.sq_hi);
        return PPF_NO_KEY;
    }
    const int kd = ppf_distance_bin(ix.tr, stocs_sqrtf(sq) * 1000.0f);   // the same float arithmetic as the reference's
    int k1, k2, k3;
    const bool c1 = ppf_angle_bin_fast(cross3(n1, u), dot3(n1, u), b, inv_rot, rot_f, &k1);
    const bool c2 = ppf_angle_bin_fast(cross3(n2, u), dot3(n2, u), b, inv_rot, rot_f, &k2);
    const bool c3 = ppf_angle_bin_fast(cross3(n1, n2), dot3(n1, n2), b, inv_rot, rot_f, &k3);
    *certain = c1 && c2 && c3;
    if ((unsigned)k1 >= (unsigned)ix.NA || (unsigned)k2 >= (unsigned)ix.NA || (unsigned)k3 >= (unsigned)ix.NA) return PPF_NO_KEY;
    return ppf_pack(kd, k1, k2, k3, ix.NA);
}
// the reference's arithmetic (double atan2) as a real call: it runs for about one pair in a thousand, and inlined its
// constants and temporaries are hoisted into registers that the surrounding loops then spill
__device__ __noinline__ uint32_t ppf_key_exact_call4(int tr, int rot, int NA, int nD, V3 p1, V3 n1, V3 p2, V3 n2) {
    int K[4];
    ppf_compute(p1, n1, p2, n2, tr, rot, K);
    IndexView ix;
    ix.bits = NULL; ix.tr = tr; ix.rot = rot; ix.NA = NA; ix.nD = nD; ix.sq_lo = ix.sq_hi = 0.0f; ix.far_certain = 1;
    return ppf_pack_or_none(ix, K);
}
__device__ __forceinline__ uint32_t ppf_key_exact_call(const IndexView& ix, V3 p1, V3 n1, V3 p2, V3 n2) {
    return ppf_key_exact_call4(ix.tr, ix.rot, ix.NA, ix.nD, p1, n1, p2, n2);
}
__device__ __forceinline__ uint32_t ppf_key_device(const IndexView& ix, V3 p1, V3 n1, V3 p2, V3 n2) {
    bool certain;
    const uint32_t key = ppf_key_fast(ix, p1, n1, p2, n2, &certain);
    return certain ? key : ppf_key_exact_call(ix, p1, n1, p2, n2);
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
    bool zero = !ppf_key_present(a.ix, ppf_key_device(a.ix, pc, nc, pi, ni)) || i == cur;
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

// Inclusive prefix sum over the 64 lanes of a wavefront with DPP row shifts and row broadcasts (no LDS traffic): lanes
// 0..15 of each row first, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {   // lanes without a source (or masked off) read 0
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, BANK_MASK, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, BANK_MASK, true);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t x) {
    uint64_t v = x + dpp_u64<0x111, 0xf, 0xf>(x) + dpp_u64<0x112, 0xf, 0xf>(x) + dpp_u64<0x113, 0xf, 0xf>(x);   // row_shr:1..3
    v += dpp_u64<0x114, 0xf, 0xe>(v);    // row_shr:4 into lanes 4..15 of each row
    v += dpp_u64<0x118, 0xf, 0xc>(v);    // row_shr:8 into lanes 8..15
    v += dpp_u64<0x142, 0xa, 0xf>(v);    // row_bcast:15 into rows 1 and 3
    v += dpp_u64<0x143, 0xc, 0xf>(v);    // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {   // l uniform
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// Seeded weighted draw over n weights by the whole 1024-thread workgroup (an exact, order-independent replacement of
// std::discrete_distribution: 2^32 fixed-point weights, r = mulhi(r64, total), first index whose inclusive prefix exceeds r;
// -1 when every weight is zero = "FAILED SAMPLING:: Zero probability returned", stocs.cpp:386-389).  Every thread sums
// consecutive weights, the wavefronts scan with DPP, every wavefront scans the wavefront totals itself: two barriers.  sh16 (2 x 16) and sh_pick (2) are double-buffered by `slot`: consecutive draws alternate it, so no
// barrier is needed behind the last read.  Returns the position in wb (or -1) to every thread.
// weight_fix (stocs_math.h) without branches: the draws run it for every weight, and divergent branches cost more than the arithmetic
__device__ __forceinline__ uint64_t weight_fix_dev(float w) {
    const uint32_t b = __float_as_uint(w);
    const int e = (int)((b >> 23) & 0xFFu);
    const uint64_t m = (uint64_t)((b & 0x7FFFFFu) | 0x800000u);
    const int sh = e - 118;
    uint64_t r = sh >= 0 ? m << min(sh, 63) : m >> min(-sh, 63);
    r = sh >= 41 ? 0xFFFFFFFFFFFFFFFFull : r;                   // 2^64 and beyond saturate (+inf included)
    r = (e == 255 && (b & 0x7FFFFFu)) ? 0 : r;                   // NaN
    r = ((int32_t)b <= 0 || e == 0) ? 0 : r;                     // negative, zero, denormal
    return r;
}

__device__ __forceinline__ int draw_block_fast(const float* __restrict__ wb, int n, uint64_t r64, uint64_t* sh16, int* sh_pick, int slot, int per_thread = 2) {
    // `per_thread` weights per thread, in as many wavefronts as that takes; the others only wait at the barriers.  Measured
    // (s_memtime inside the instance kernel): a wavefront-level step costs its instructions x 4 cycles x the wavefronts
    // sharing a SIMD, and the winner's rescan of its own weights is serial, so short chunks win despite the fixed scans.
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nt = min((int)blockDim.x, max(64, (((n + per_thread - 1) / per_thread) + 63) & ~63)), nw = nt >> 6;   // (blockDim.x: 1024, or 512 in the lean kernel's small form)
    const int chunk = (n + nt - 1) / nt;
    const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
    uint64_t local = 0, incl = 0;
    if (t < nt) {
        for (int i = lo; i < hi; i += 4) {                      // loads first, four at a time
            const float v0 = wb[i], v1 = i + 1 < hi ? wb[i + 1] : 0.0f, v2 = i + 2 < hi ? wb[i + 2] : 0.0f, v3 = i + 3 < hi ? wb[i + 3] : 0.0f;
            local += (weight_fix_dev(v0) + weight_fix_dev(v1)) + (weight_fix_dev(v2) + weight_fix_dev(v3));
        }
            incl = wave_incl_scan_u64(local);
        if (lane == 63) sh16[slot * 16 + wv] = incl;
    }
    __syncthreads();
    if (t < nt) {
        const uint64_t tv = lane < nw ? sh16[slot * 16 + lane] : 0;
        const uint64_t tinc = wave_incl_scan_u64(tv);           // lanes 0..nw-1: inclusive scan of the wavefront totals
        const uint64_t total = readlane_u64(tinc, nw - 1);
        if (total == 0) { if (t == 0) sh_pick[slot] = -1; }
        else {
            const uint64_t r = mulhi64(r64, total);
            const uint64_t in2 = readlane_u64(tinc - tv, wv) + incl, ex2 = in2 - local;
            if (r >= ex2 && r < in2) {                          // exactly one thread (its local > 0)
                uint64_t c = ex2;
                int pick = -1;
                for (int i = lo; i < hi && pick < 0; i += 4) {
                    const float v0 = wb[i], v1 = i + 1 < hi ? wb[i + 1] : 0.0f, v2 = i + 2 < hi ? wb[i + 2] : 0.0f, v3 = i + 3 < hi ? wb[i + 3] : 0.0f;
                    const uint64_t c0 = c + weight_fix_dev(v0), c1 = c0 + weight_fix_dev(v1), c2 = c1 + weight_fix_dev(v2), c3 = c2 + weight_fix_dev(v3);
                    pick = c0 > r ? i : (c1 > r ? i + 1 : (c2 > r ? i + 2 : (c3 > r ? i + 3 : -1)));
                    c = c3;
                }
                sh_pick[slot] = pick;
            }
        }
    }
    __syncthreads();
    return sh_pick[slot];
}

// one workgroup per attempt
__global__ __launch_bounds__(1024) void draw_kernel(const float* __restrict__ w, size_t stride, int S, uint64_t seed,
                                                    uint64_t first_attempt, uint64_t k, const uint64_t* __restrict__ r_explicit,
                                                    int slot, int32_t* __restrict__ bidx, int32_t* __restrict__ fail) {
    __shared__ uint64_t sh16[32];
    __shared__ int sh_pick[2];
    const int b = blockIdx.x;
    if (fail[b]) return;
    const uint64_t r64 = r_explicit ? r_explicit[b] : rng64(seed, first_attempt + (uint64_t)b, k);
    const int pick = draw_block_fast(w + (size_t)b * stride, S, r64, sh16, sh_pick, 0);
    if (threadIdx.x == 0) {
        bidx[b * 4 + slot] = pick;
        if (pick < 0) fail[b] = 1;
    }
}

// self-check of weight_fix_dev against the double formula, 2^32 patterns: one thread per 256 consecutive patterns
__global__ __launch_bounds__(256) void weight_fix_check_kernel(unsigned int* __restrict__ counts) {
    const uint32_t first = (blockIdx.x * 256u + threadIdx.x) << 8;
    unsigned bad = 0;
    for (uint32_t k = 0; k < 256; ++k) {
        const float w = __uint_as_float(first + k);
        uint64_t ref = 0;
        if (w > 0.0f) {
            const double s = (double)w * 4294967296.0;
            ref = s >= 1.8446744073709552e19 ? 0xFFFFFFFFFFFFFFFFull : (uint64_t)s;
        }
        bad += (weight_fix_dev(w) != ref || weight_fix(w) != ref) ? 1u : 0u;
    }
    if (bad) atomicAdd(&counts[0], bad);
}

// self-check of the filter on the context's own scene: pair (i, j) of every thread with both evaluations
__global__ __launch_bounds__(256) void ppf_filter_check_kernel(PassArgs a, uint64_t seed, uint32_t n_pairs, unsigned int* __restrict__ counts) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_pairs) return;
    const uint64_t r = rng64(seed, e, 7);
    const int i = (int)((r & 0xFFFFFFFFull) % (uint64_t)a.S), j = (int)((r >> 32) % (uint64_t)a.S);
    if (i == j) return;
    const float4 p1 = a.spos[i], n1 = a.snrm[i], p2 = a.spos[j], n2 = a.snrm[j];
    const V3 P1 = mk3(p1.x, p1.y, p1.z), N1 = mk3(n1.x, n1.y, n1.z), P2 = mk3(p2.x, p2.y, p2.z), N2 = mk3(n2.x, n2.y, n2.z);
    bool certain;
    const uint32_t kf = ppf_key_fast(a.ix, P1, N1, P2, N2, &certain);
    int Ke[4];
    ppf_compute(P1, N1, P2, N2, a.ix.tr, a.ix.rot, Ke);
    const uint32_t ke = ppf_pack_or_none(a.ix, Ke);
    atomicAdd(&counts[0], 1u);
    if (!certain) atomicAdd(&counts[1], 1u);
    if (certain && kf != ke) atomicAdd(&counts[2], 1u);          // a pair the filter called certain and got wrong
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
// (struct BaseOut: stocs_ctx.h)

// ordered base + invariants of one attempt by twelve lanes of one wavefront: the 12 ordered pairings of try_sampled_base
// (stocs.cpp:224-268) evaluated side by side, then the reference's first-strict-minimum rule in its enumeration order.
// Callers pass a whole wavefront.
__device__ __forceinline__ void finalize_one_wave(const float4* __restrict__ spos, const int32_t* bidx4, int fail, BaseOut* o) {
    const int lane = threadIdx.x & 63;
    int ids[4] = {bidx4[0], bidx4[1], bidx4[2], bidx4[3]};
    const bool ok0 = !fail && ids[0] >= 0 && ids[1] >= 0 && ids[2] >= 0 && ids[3] >= 0;
    float dist = 3.402823466e+38f, fi1 = 0, fi2 = 0;
    int pi = 0, pj = 1, pk = 2, pl = 3;
    if (ok0 && lane < 12) {
        // pairing number lane in the order of the loops of stocs.cpp:230-257: i outer, j != i inner
        pi = lane / 3;
        const int jj = lane % 3;
        pj = jj + (jj >= pi ? 1 : 0);
        pk = 0; while (pk == pi || pk == pj) pk++;
        pl = 0; while (pl == pi || pl == pj || pl == pk) pl++;
        V3 base[4];
        for (int k = 0; k < 4; ++k) { const float4 p = spos[ids[k]]; base[k] = mk3(p.x, p.y, p.z); }
        double li1, li2;
        dist = (float)seg_dist_inv(base[pi], base[pj], base[pk], base[pl], li1, li2);
        fi1 = (float)li1; fi2 = (float)li2;
    }
    // first strict minimum in enumeration order == smallest distance, lowest lane among equals (NaN distances never win)
    float best = dist;
    for (int off = 8; off > 0; off >>= 1) best = fminf(best, __shfl_xor(best, off, 64));   // lanes 0..15 hold the 12 values + max floats
    const unsigned long long m = __ballot(lane < 12 && dist == best && dist < 3.402823466e+38f);
    const int win = m ? __ffsll((long long)m) - 1 : -1;
    if (lane == (win < 0 ? 0 : win)) {
        const bool ok = ok0 && win >= 0;
        const int tmp[4] = {ids[0], ids[1], ids[2], ids[3]};
        if (ok) { ids[0] = tmp[pi]; ids[1] = tmp[pj]; ids[2] = tmp[pk]; ids[3] = tmp[pl]; }
        for (int k = 0; k < 4; ++k) o->ids[k] = ids[k];
        o->inv[0] = ok ? fi1 : 0.0f; o->inv[1] = ok ? fi2 : 0.0f;
        o->valid = ok ? 1 : 0;
        o->pad = 0;
    }
}

__global__ __launch_bounds__(64) void finalize_bases_kernel(const float4* __restrict__ spos, const int32_t* __restrict__ bidx,
                                                            const int32_t* __restrict__ fail, int nB, BaseOut* __restrict__ out) {
    const int b = blockIdx.x;   // one wavefront per attempt, the 12 pairings side by side
    if (b < nB) finalize_one_wave(spos, bidx + 4 * b, fail[b], out + b);
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

// Distance component of the key in terms of the squared length.  With v = int(|u| * 1000) (rgbd.cpp:103) the bin index is
// kd = floor((v - b + tr) / tr), b = ceil(tr / 2); a key exists only for kd * tr > 5 and kd < nD, i.e. v_lo <= v < v_hi.
// sqrt (correctly rounded) and the float product are monotone, so that is sq_lo <= |u|^2 < sq_hi for two floats found by
// bisection with the very float operations of the kernels (stocs_sqrtf, one float multiply).
static void set_distance_thresholds(IndexView* ix) {
    const int tr = ix->tr, b = (tr + 1) / 2;
    const long long v_lo = (long long)(5 / tr + 1) * tr + b - tr, v_hi_full = (long long)ix->nD * tr + b - tr;
    const long long v_hi = std::min<long long>(v_hi_full, 1 << 20);
    ix->far_certain = v_hi_full <= (1 << 20) ? 1 : 0;
    auto first_sq = [](float target) {   // smallest non-negative float q with sqrt(q) * 1000 >= target (+inf when there is none)
        auto pred = [target](uint32_t bits) { float q; memcpy(&q, &bits, 4); return stocs_sqrtf(q) * 1000.0f >= target; };
        uint32_t lo = 0, hi = 0x7F800000u;   // +0 .. +inf: the bit patterns of the non-negative floats are ordered
        if (pred(lo)) return 0.0f;
        while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (pred(mid)) hi = mid; else lo = mid; }
        float q; memcpy(&q, &hi, 4);
        return q;
    };
    ix->sq_lo = first_sq((float)v_lo);
    ix->sq_hi = v_hi > v_lo ? first_sq((float)v_hi) : ix->sq_lo;
}

static PassArgs pass_args(const stocs_ctx* c) {
    PassArgs a;
    a.spos = c->d_spos; a.snrm = c->d_snrmw; a.S = c->nS;
    a.ix.bits = c->index.d_exists; a.ix.tr = c->index.tr; a.ix.rot = c->index.rot; a.ix.NA = c->index.NA; a.ix.nD = c->index.nD;
    a.ix.sq_lo = 0.0f; a.ix.sq_hi = 0.0f; a.ix.far_certain = 1;
    if (a.ix.tr > 0) set_distance_thresholds(&a.ix);
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

static int sample_class_multi(stocs_ctx* c, uint64_t seed, int first_attempt, int nB, int32_t* ids, float* inv, int32_t* valid) {
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
    hipLaunchKernelGGL(finalize_bases_kernel, dim3((unsigned)nB), dim3(64), 0, c->stream, c->d_spos, sb.bidx, sb.fail, nB, sb.res);
    STOCS_HIP_CHECK(hipGetLastError());
    std::vector<BaseOut> res((size_t)nB);
    STOCS_HIP_CHECK(hipMemcpyAsync(res.data(), sb.res, (size_t)nB * sizeof(BaseOut), hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return record_bases(c, nB, res.data(), ids, inv, valid);
}

// ---------------------------------------------------------------------------------------------------------------
// Instance mode (stocs.cpp:559-751) entirely on the device: ONE persistent 1024-thread workgroup runs the attempts one
// after the other -- they are sequential by construction (compounding prior decay :572-580, previous_segment :626, the
// segmentation buffer) -- with workgroup barriers between the stages and no host involvement until the results are read.
//
// Image-space state per SCENE POINT instead of per pixel: everything the reference reads from previous_segment /
// segmentation_buffer / seg_mask_<n>.png (rgbd.cpp:314-367, Q14) it reads at the pixel of a scene point, so the state is
//   prev_in[i]      pixel of point i lies in previous_segment
//   label[i]        segmentation_buffer at that pixel (attempt that last flood-filled it, 0 = none)
//   maskbits[n][i]  pixel of point i lies in attempt n's mask (the seg_mask_<n>.png read back at rgbd.cpp:322-326)
// The flood fill itself (rgbd.cpp:334-366: 8-neighbourhood over passable pixels -- png value 255 -- closer to the seed
// than max_distance) is a connected-component query: the passable pixels of every image row are kept as runs (built once
// per edge map), the runs of the rows inside the disc are clipped to it and united when they touch (lock-free union-find
// in LDS), and a point is in the mask when its run has the seed's root.  sqrt(d2) < sqrt(max d2) in float is decided on
// the integers d2 (distinct integers below 2^20 have distinct float square roots).
// ---------------------------------------------------------------------------------------------------------------
#define INST_MAX_NODES 16384
// the whole per-point working state of an attempt (weights 4 B + survivor index 2 B per point) lives in LDS next to the
// union-find parents up to this many scene points; larger scenes keep it in device memory (same code, other pointers)
#define INST_LDS_POINTS 16000

// two passable runs of consecutive image rows that touch in the 8-neighbourhood (unclipped): the edges of the run graph,
// built once per edge map.  g = run in `row`, g + dh = run in row + 1; [gs, ge) / [hs, he) their columns.
struct RunPair {
    uint32_t g;
    uint16_t dh, row;
    uint16_t gs, ge, hs, he;
};

struct InstanceArgs {
    PassArgs pa;
    const int2* pix;            // (row, col) per scene point
    const uint8_t* edge_pt;     // point's pixel has edge_probability == 1 (png value 0): pruned (stocs.cpp:521-535)
    const int32_t* pt_run;      // run holding the point's pixel, -1 when the pixel is not passable
    const uint16_t* run_s; const uint16_t* run_e;   // [start, end) columns of the passable runs, row by row
    const uint32_t* row_off;    // H + 1
    const RunPair* pairs;       // touching runs of rows (r, r + 1), row by row
    const uint32_t* pair_off;   // H + 1: pairs of rows (r, r + 1) start at pair_off[r]
    int H, W, Sw;               // Sw = words per point bitset
    float* cls;                 // current class probabilities (decay in place, Q8)
    uint8_t* prev_in; uint8_t* label;
    uint32_t* maskbits;         // 256 x Sw
    uint32_t* segbits;          // Sw: `segment` of the last attempt
    uint32_t* parent_g;         // union-find parents when the disc holds more than INST_MAX_NODES runs (+ 1)
    float* w;                   // S weights (device-memory variant only)
    int32_t* sv;                // S survivor indices (device-memory variant only)
    int draw_per_thread;        // weights per thread the draws aim at
    unsigned long long* stamps; // STOCS_DEBUG_TIMING only: cycles per stage, summed over the attempts (else NULL)
    float4* spos_w; float4* snrm_w;   // the scene arrays whose .w the LCP adds: refreshed with the decayed prior at the end
    BaseOut* res;
    // hand-over from the workgroup that runs point 1 + mask of every attempt to the one that runs points 2..4 (below)
    int4* q_hdr;                // per attempt: survivors, point 1, 1 when the attempt got as far as its mask
    int32_t* q_sv; float* q_w;  // per attempt S slots: the survivors (scene index, weight)
    unsigned int* q_flag;       // per attempt: 1 once the slot is complete
    unsigned int* q_err;        // set when the second workgroup gave up waiting
    // trial batches (stocs_run_trials): workgroups 2 t and 2 t + 1 are the two roles of trial t.  Every pointer from `cls` down
    // (except spos_w) then names trial 0's copy inside a block of per-trial state, and trial t's copy lies t * trial_stride bytes
    // behind it; seeds[t] is the trial's seed.
    int n_trials;               // 0: one trial on the context's own state (the kernel's seed argument)
    size_t trial_stride;
    const uint64_t* seeds;
};

template <class P>
__device__ __forceinline__ uint32_t uf_find(P parent, uint32_t x) {
    uint32_t p = parent[x];
    while (p != x) { const uint32_t g = parent[p]; parent[x] = g; x = p; p = g; }   // path halving; races only ever shorten paths
    return x;
}
template <class P>
__device__ __forceinline__ void uf_unite(P parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a); b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }      // the larger root goes under the smaller one
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}
// largest h >= 0 with h*h < rem (rem > 0)
__device__ __forceinline__ int half_width(int rem) {
    int h = (int)sqrtf((float)rem);
    while (h * h >= rem) --h;
    while ((h + 1) * (h + 1) < rem) ++h;
    return h;
}
// columns [s, e) of a run of row r clipped to the open disc d2 < maxd2 around (r0, c0); empty when *cs >= *ce
__device__ __forceinline__ void clip_run(int s, int e, int r, int r0, int c0, int maxd2, int* cs, int* ce) {
    const int dr = r - r0, rem = maxd2 - dr * dr;
    if (rem <= 0) { *cs = 0; *ce = 0; return; }
    const int hw = half_width(rem);
    *cs = max(s, c0 - hw);
    *ce = min(e, c0 + hw + 1);
}

// connected component of the seed (rgbd.cpp:334-366): the run pairs of the rows inside the disc are clipped and united when
// the clipped intervals still touch; the seed pixel (node `nodes`) touches every clipped run that meets its 3x3
// neighbourhood.  Afterwards a point is inside the mask when its run has the seed's root.
template <class ParentPtr>
__device__ void flood_fill_runs(const InstanceArgs& A, ParentPtr parent, int r0, int c0, int maxd2, int rlo, int rhi, uint32_t base, uint32_t nodes) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t p0 = A.pair_off[rlo], p1 = A.pair_off[rhi];
    for (uint32_t k = t; k <= nodes; k += 1024) parent[k] = k;
    __syncthreads();
    for (uint32_t pb = p0; pb < p1; pb += 4096) {
        uint4 raw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t p = pb + k * 1024 + t; raw[k] = make_uint4(0, 0, 0, 0); if (p < p1) raw[k] = ((const uint4*)A.pairs)[p]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = pb + k * 1024 + t;
            if (p >= p1) continue;
            const uint32_t g = raw[k].x;
            const int dh = (int)(raw[k].y & 0xFFFFu), row = (int)(raw[k].y >> 16);
            int cs, ce, ds, de;
            clip_run((int)(raw[k].z & 0xFFFFu), (int)(raw[k].z >> 16), row, r0, c0, maxd2, &cs, &ce);
            if (cs >= ce) continue;
            clip_run((int)(raw[k].w & 0xFFFFu), (int)(raw[k].w >> 16), row + 1, r0, c0, maxd2, &ds, &de);
            if (ds < de && ds < ce + 1 && de > cs - 1) uf_unite(parent, g - base, g + (uint32_t)dh - base);   // 8-neighbourhood: overlap widened by one
        }
    }
    if (wv < 3) {                                              // one wavefront per row of the seed's 3x3 neighbourhood
        const int r = r0 - 1 + wv;
        if (r >= rlo && r <= rhi) {
            const uint32_t a0 = A.row_off[r], a1 = A.row_off[r + 1];
            for (uint32_t g = a0 + lane; g < a1; g += 64) {
                int cs, ce;
                clip_run((int)A.run_s[g], (int)A.run_e[g], r, r0, c0, maxd2, &cs, &ce);
                if (cs < ce && cs < c0 + 2 && ce > c0 - 1) uf_unite(parent, g - base, nodes);
            }
        }
    }
    __syncthreads();
}

__device__ __noinline__ void finalize_one_wave_call(const float4* spos, int b0, int b1, int b2, int b3, int fail, BaseOut* o) {
    const int32_t bidx[4] = {b0, b1, b2, b3};
    finalize_one_wave(spos, bidx, fail, o);
}

template <bool WLDS> struct InstTypes { typedef int32_t sv_t; };
template <> struct InstTypes<true> { typedef uint16_t sv_t; };

// Every loop over the scene points handles 4 points per thread with the loads of all four issued before the first use:
// the stages run in one workgroup, so memory latency is not hidden by other workgroups -- it is paid once per stage instead
// of once per point.
//
// TWO workgroups share the attempts as a pipeline.  What makes the attempts sequential is the image-space state (the
// decayed prior, previous_segment, the segmentation buffer): it is complete once an attempt has its mask.  Points 2..4 of
// an attempt (stocs.cpp:640-751) read that attempt's survivors only and write nothing a later attempt reads.  So the
// first workgroup of the grid runs weights -> point 1 -> pass 1 -> mask -> bookkeeping of every attempt and hands the
// survivors over through device memory (slot + agent-scope release flag); the second workgroup picks the slots up
// (acquire), draws points 2..4 and orders the bases.  The two normally sit on neighbouring XCDs with separate L2s;
// placing them on one XCD (a grid of 9 with 7 idle workgroups, since workgroups are dealt to the XCDs in turn) was
// measured and made no difference (1.81 against 1.82 ms per 100 attempts), so the grid is just the two.  The second
// workgroup bounds its wait, so a failure of the first cannot hang the device.
template <bool WLDS>
__global__ __launch_bounds__(1024) void instance_attempts_kernel(InstanceArgs A, uint64_t seed, int first_attempt, int n_attempts, float dispersion) {
    typedef typename InstTypes<WLDS>::sv_t sv_t;
    extern __shared__ __align__(16) unsigned char inst_dyn[];
    __shared__ uint64_t sh16[32];
    __shared__ int sh_pick[2];
    __shared__ int sh_max, sh_nunc;
    __shared__ int sh_cnt[64], sh_cex[65];
    uint32_t* parent_l = (uint32_t*)inst_dyn;                               // INST_MAX_NODES + 1
    const int S = A.pa.S;
    const size_t o_w = ((size_t)(INST_MAX_NODES + 1) * 4 + 15) & ~(size_t)15;
    float* w = WLDS ? (float*)(inst_dyn + o_w) : A.w;                       // weights; after the compaction the survivors' weights
    sv_t* sv = WLDS ? (sv_t*)(inst_dyn + o_w + (((size_t)S * 4 + 15) & ~(size_t)15)) : (sv_t*)A.sv;   // survivors of pass 1 inside the mask, in scene order
    const float4* spos = A.pa.spos;
    const float4* snrm = A.pa.snrm;
    bool first_role = blockIdx.x == 0;
    if (A.n_trials > 0) {
        const int trial = (int)(blockIdx.x >> 1);
        if (trial >= A.n_trials) return;
        first_role = (blockIdx.x & 1u) == 0u;
        seed = A.seeds[trial];
        const size_t off = (size_t)trial * A.trial_stride;
#define INST_ADV(p) p = (decltype(p))((char*)(p) + off)
        INST_ADV(A.cls); INST_ADV(A.prev_in); INST_ADV(A.label); INST_ADV(A.maskbits); INST_ADV(A.segbits); INST_ADV(A.parent_g); INST_ADV(A.w); INST_ADV(A.sv);
        INST_ADV(A.snrm_w); INST_ADV(A.res); INST_ADV(A.q_hdr); INST_ADV(A.q_sv); INST_ADV(A.q_w); INST_ADV(A.q_flag); INST_ADV(A.q_err);
#undef INST_ADV
    } else if (blockIdx.x != 0 && blockIdx.x + 1 != gridDim.x) return;     // (placement experiments launch idle workgroups in between)
    unsigned long long tprev = A.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    // the thread index is made opaque at the start of every stage: otherwise the compiler hoists per-thread addresses of
    // every array out of the attempt loop, and a hundred registers of them spill to scratch around every stage
#define INST_THREAD() int t = threadIdx.x; asm volatile("" : "+v"(t)); const int lane = t & 63, wv = t >> 6; (void)lane; (void)wv;
    // (debug) the stage clocks are summed in scalar registers and written once at the end: a read-modify-write of device
    // memory per stamp would cost more than most stages
    unsigned long long acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0, acc5 = 0, acc6 = 0, acc7 = 0;
#define INST_STAMP(k) if (A.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc##k += now_ - tprev; tprev = now_; }
    if (first_role) {
    for (int a = 0; a < n_attempts; ++a) {
        const int attempt = first_attempt + a, base_num = attempt + 1;
        // ---- weights: compounding decay of the prior inside the previous segment, edge pixels pruned (stocs.cpp:572-584) ----
        {
        INST_THREAD()
        for (int i0 = 0; i0 < S; i0 += 4096) {
            float c[4]; uint8_t pin[4], ep[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * 1024 + t;
                c[k] = 0.0f; pin[k] = 0; ep[k] = 0;
                if (i < S) { c[k] = A.cls[i]; pin[k] = A.prev_in[i]; ep[k] = A.edge_pt[i]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * 1024 + t;
                if (i >= S) continue;
                float ck = c[k];
                if (pin[k]) { ck = dispersion * ck; A.cls[i] = ck; }
                w[i] = ep[k] ? 0.0f : ck;
            }
        }
        if (t == 0) { sh_max = 0; sh_nunc = 0; }
        }
        __syncthreads();
        // every store to the slot of the previous attempt has completed (the barrier waits for them): publish it
        if (a > 0 && threadIdx.x == 0) __hip_atomic_store(A.q_flag + (a - 1), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        INST_STAMP(0)
        int32_t bidx[4] = {-1, -1, -1, -1};
        bidx[0] = draw_block_fast(w, S, rng64(seed, (uint64_t)attempt, 0), sh16, sh_pick, 0, A.draw_per_thread);
        INST_STAMP(1)
        if (bidx[0] < 0) {   // "FAILED SAMPLING": no base, no mask, previous_segment stays (stocs.cpp:586-589)
            if (threadIdx.x == 0) A.q_hdr[a] = make_int4(0, -1, 0, 0);
            continue;
        }
        const int b1 = bidx[0];
        const int2 sp = A.pix[b1];
        const int lab = A.label[b1];
        // ---- pass 1 (stocs.cpp:596-609) + the largest pixel distance of a survivor (:610-618) ----
        {
            INST_THREAD()
            const float4 pc4 = spos[b1], nc4 = snrm[b1];
            const V3 pc = mk3(pc4.x, pc4.y, pc4.z), nc = mk3(nc4.x, nc4.y, nc4.z);
            int my_max = 0;
            // stage A, distance alone (the first component of the key, rgbd.cpp:103): a point farther from P1 than the
            // longest model pair (or within 5 mm) has no key whatever its angles; the others are listed in `sv`
            for (int i0 = 0; i0 < S; i0 += 4096) {
                float wi[4]; float4 P[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * 1024 + t;
                    wi[k] = (i < S) ? w[i] : 0.0f;
                    P[k] = make_float4(0, 0, 0, 0);
                    if (wi[k] != 0.0f) P[k] = spos[i];
                }
                bool cand[4];
                unsigned long long cm[4];
                int n_here = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * 1024 + t;
                    cand[k] = false;
                    if (wi[k] != 0.0f) {
                        cand[k] = i != b1 && ppf_distance_may_have_key(A.pa.ix, pc - mk3(P[k].x, P[k].y, P[k].z));
                        if (!cand[k]) w[i] = 0.0f;
                    }
                    cm[k] = __ballot(cand[k]);
                    n_here += __popcll(cm[k]);
                }
                int base_pos = 0;                                             // one atomic per wavefront for its (up to) 256 points
                if (lane == 0 && n_here) base_pos = atomicAdd(&sh_nunc, n_here);
                base_pos = __builtin_amdgcn_readfirstlane(base_pos);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (cand[k]) sv[base_pos + __popcll(cm[k] & ((1ull << lane) - 1ull))] = (sv_t)(i0 + k * 1024 + t);
                    base_pos += __popcll(cm[k]);
                }
            }
            __syncthreads();
            // stage B, the angles of the listed points (two per thread and turn, loads first)
            const int n_cand = sh_nunc;
            if (A.stamps && threadIdx.x == 0) { A.stamps[8] += (unsigned long long)n_cand; A.stamps[10] += __builtin_amdgcn_s_memtime() - tprev; }
            for (int j0 = 0; j0 < n_cand; j0 += 2048) {
                int ii[2]; float4 P[2], N[2]; int2 px[2]; uint32_t key[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = j0 + k * 1024 + t;
                    ii[k] = -1; P[k] = make_float4(0, 0, 0, 0); N[k] = P[k]; px[k] = make_int2(0, 0);
                    if (j < n_cand) { ii[k] = (int)sv[j]; P[k] = spos[ii[k]]; N[k] = snrm[ii[k]]; px[k] = A.pix[ii[k]]; }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    key[k] = PPF_NO_KEY;
                    if (ii[k] >= 0) key[k] = ppf_key_device(A.pa.ix, pc, nc, mk3(P[k].x, P[k].y, P[k].z), mk3(N[k].x, N[k].y, N[k].z));
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (ii[k] < 0) continue;
                    if (!ppf_key_present(A.pa.ix, key[k])) w[ii[k]] = 0.0f;
                    else { const int dr = sp.x - px[k].x, dc = sp.y - px[k].y; my_max = max(my_max, dr * dr + dc * dc); }
                }
            }
            for (int off = 32; off > 0; off >>= 1) my_max = max(my_max, __shfl_xor(my_max, off, 64));
            if (lane == 0 && my_max) atomicMax(&sh_max, my_max);
        }
        __syncthreads();
        INST_STAMP(2)
        const int maxd2 = sh_max;
        // ---- the mask: an earlier attempt's when the seed pixel is already labelled, else a new flood fill (rgbd.cpp:314-367) ----
        const int rad = maxd2 > 0 ? half_width(maxd2) : 0;
        const int rlo = max(0, sp.x - rad), rhi = min(A.H - 1, sp.x + rad);
        const uint32_t base = A.row_off[rlo], nodes = A.row_off[rhi + 1] - base;
        const bool in_lds = nodes <= INST_MAX_NODES;
        uint32_t root_s = 0;
        if (lab == 0) {
            if (in_lds) flood_fill_runs(A, parent_l, sp.x, sp.y, maxd2, rlo, rhi, base, nodes);
            else flood_fill_runs(A, A.parent_g, sp.x, sp.y, maxd2, rlo, rhi, base, nodes);
            root_s = in_lds ? uf_find(parent_l, nodes) : uf_find(A.parent_g, nodes);
        }
        INST_STAMP(3)
        // ---- bookkeeping per scene point: mask membership, previous_segment, labels, and the survivors inside the mask
        //      (`segment`, stocs.cpp:628-638) compacted in scene order (in place: position <= index): points 2..4 are
        //      drawn among them alone ----
        int n_surv = 0;
        {
        INST_THREAD()
        for (int i0 = 0; i0 < S; i0 += 4096) {
            float wi[4]; int2 px[4]; int run[4]; uint32_t mw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * 1024 + t;
                wi[k] = 0.0f; run[k] = -1; mw[k] = 0; px[k] = make_int2(0, 0);
                if (i < S) {
                    wi[k] = w[i];
                    if (lab != 0) mw[k] = A.maskbits[(size_t)lab * A.Sw + (i >> 5)];
                    else { px[k] = A.pix[i]; run[k] = A.pt_run[i]; }
                }
            }
            unsigned long long sbal[4];
            bool keep[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                     // whole wavefronts: the bitset words are ballots
                const int i = i0 + k * 1024 + t;
                bool in = false;
                if (i < S) {
                    if (lab != 0) in = (mw[k] >> (i & 31)) & 1u;
                    else {
                        const int dr = px[k].x - sp.x, dc = px[k].y - sp.y;
                        if (dr == 0 && dc == 0) in = true;                    // the seed pixel is always part of its mask
                        else if (dr * dr + dc * dc < maxd2 && run[k] >= 0)
                            in = (in_lds ? uf_find(parent_l, (uint32_t)run[k] - base) : uf_find(A.parent_g, (uint32_t)run[k] - base)) == root_s;
                    }
                    A.prev_in[i] = in ? 1 : 0;                                // segmentation_mask.copyTo(previous_segment), stocs.cpp:626
                    if (lab == 0 && in) A.label[i] = (uint8_t)base_num;       // segmentation_buffer = base_num over the new mask
                }
                keep[k] = in && wi[k] != 0.0f;
                const unsigned long long mb = __ballot(in);
                sbal[k] = __ballot(keep[k]);
                if (lane == 0) {
                    const int wd = ((i0 + k * 1024) >> 5) + 2 * wv;
                    if (wd < A.Sw) {
                        A.maskbits[(size_t)base_num * A.Sw + wd] = (uint32_t)mb;   // seg_mask_<base_num>.png, whichever mask it is
                        A.segbits[wd] = (uint32_t)sbal[k];
                        A.maskbits[(size_t)base_num * A.Sw + wd + 1] = (uint32_t)(mb >> 32); A.segbits[wd + 1] = (uint32_t)(sbal[k] >> 32);
                    }
                    sh_cnt[k * 16 + wv] = __popcll(sbal[k]);                   // scene order inside the block of 4096: k, wavefront, lane
                }
            }
            __syncthreads();
            if (wv == 0) {
                const int v = sh_cnt[lane];
                int inc = v;
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
                sh_cex[lane] = inc - v;
                if (lane == 63) sh_cex[64] = inc;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (keep[k]) {
                    const int pos = n_surv + sh_cex[k * 16 + wv] + __popcll(sbal[k] & ((1ull << lane) - 1ull));
                    sv[pos] = (sv_t)(i0 + k * 1024 + t); w[pos] = wi[k];
                }
            n_surv += sh_cex[64];
            __syncthreads();
        }
        }
        // ---- hand the survivors over (published at the next barrier that follows these stores) ----
        {
            INST_THREAD()
            for (int j = t; j < n_surv; j += 1024) { A.q_sv[(size_t)a * S + j] = (int32_t)sv[j]; A.q_w[(size_t)a * S + j] = w[j]; }
            if (t == 0) A.q_hdr[a] = make_int4(n_surv, b1, 1, 0);
        }
        INST_STAMP(4)
        if (A.stamps && threadIdx.x == 0) A.stamps[9] += (unsigned long long)n_surv;
    }
    __syncthreads();
    if (n_attempts > 0 && threadIdx.x == 0) __hip_atomic_store(A.q_flag + (n_attempts - 1), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (A.stamps && threadIdx.x == 0) { A.stamps[0] += acc0; A.stamps[1] += acc1; A.stamps[2] += acc2; A.stamps[3] += acc3; A.stamps[4] += acc4; }
    // the LCP adds class_probability_, which this sampling decays in place (Q8): refresh the scene arrays it reads
    for (int i = threadIdx.x; i < S; i += 1024) { const float c = A.cls[i]; if (A.spos_w) A.spos_w[i].w = c; A.snrm_w[i].w = c; }
    return;
    }

    // ================= second workgroup: points 2..4 (stocs.cpp:640-751 = the class-mode passes) over the survivors =================
    __shared__ int sh_abort;
    if (threadIdx.x == 0) sh_abort = 0;
    __syncthreads();
    for (int a = 0; a < n_attempts; ++a) {
        const int attempt = first_attempt + a;
        BaseOut* out = A.res + a;
        if (threadIdx.x == 0) {
            // Wait for the slot.  The producer of this pair has the lower workgroup index, so it was dispatched BEFORE this workgroup
            // and is resident (or done) whatever the number of CUs, partitions or other contexts on the device: it never waits for
            // anybody, so a busy device can only delay it.  The bound is therefore in wall-clock time of the constant 100 MHz counter
            // (20 s: an exit condition every wave reaches, far beyond any delay a shared device produces), not in polls -- round 4
            // counted 2^21 polls, ~2-4 s, which another context's 1024-thread workgroups on the same CUs could have exhausted.
            // Relaxed polls (a coherent read of the one flag word); the acquire is the fence every thread executes behind the
            // barrier below -- an acquire per poll would invalidate this XCD's L2 a million times a second under whoever shares it
            const unsigned long long t_wait0 = wall_clock64();
            while (__hip_atomic_load(A.q_flag + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t_wait0 > 2000000000ull) { sh_abort = 1; break; }
            }
        }
        __syncthreads();
        if (sh_abort) {                                                       // the same in every thread
            if (threadIdx.x == 0) {
                *A.q_err = 1u;
                for (int r = a; r < n_attempts; ++r) { BaseOut* o = A.res + r; for (int k = 0; k < 4; ++k) o->ids[k] = -1; o->inv[0] = o->inv[1] = 0; o->valid = 0; o->pad = 0; }
            }
            break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                    // every thread reads the slot behind the flag
        INST_STAMP(5)
        const int4 hdr = A.q_hdr[a];
        int32_t bidx[4] = {hdr.y, -1, -1, -1};
        int fail = hdr.z ? 0 : 1;
        const int n_surv = hdr.x;
        float* wq = w;
        sv_t* svq = sv;
        if (!fail) {
            INST_THREAD()
            if (WLDS) {
                for (int j = t; j < n_surv; j += 1024) { w[j] = A.q_w[(size_t)a * S + j]; sv[j] = (sv_t)A.q_sv[(size_t)a * S + j]; }
                __syncthreads();
            } else {                                                          // large scenes: work on the slot itself
                wq = A.q_w + (size_t)a * S;
                svq = (sv_t*)(A.q_sv + (size_t)a * S);
            }
        }
        for (int k = 1; k < 4 && !fail; ++k) {
            INST_THREAD()
            const int pos = draw_block_fast(wq, n_surv, rng64(seed, (uint64_t)attempt, (uint64_t)k), sh16, sh_pick, k & 1, A.draw_per_thread);
            if (pos < 0) { fail = 1; break; }
            bidx[k] = (int32_t)svq[pos];
            if (k < 3) {
                for (int j = t; j < n_surv; j += 1024) {
                    if (wq[j] == 0.0f) continue;                              // already zero: nothing to decide
                    const int i = (int)svq[j];
                    const bool z = (k == 1) ? pass_zeroes<2>(A.pa, bidx[0], bidx[1], -1, i) : pass_zeroes<3>(A.pa, bidx[0], bidx[1], bidx[2], i);
                    if (z) wq[j] = 0.0f;
                }
                __syncthreads();
            }
        }
        // the four points as drawn; their ordering (try_sampled_base) waits for the end of the kernel
        if (threadIdx.x == 0) { for (int k = 0; k < 4; ++k) out->ids[k] = bidx[k]; out->inv[0] = out->inv[1] = 0; out->valid = fail ? 0 : 1; out->pad = 0; }
        __syncthreads();
        INST_STAMP(6)
    }
    __syncthreads();
    // ---- ordered base + invariants of every attempt (stocs.cpp:224-268), side by side, one wavefront each ----
    if (!sh_abort)
        for (int a = (int)(threadIdx.x >> 6); a < n_attempts; a += 16) {
            BaseOut* o = A.res + a;
            const int b0 = o->ids[0], b1 = o->ids[1], b2 = o->ids[2], b3 = o->ids[3], fl = o->valid ? 0 : 1;
            finalize_one_wave_call(A.pa.spos, b0, b1, b2, b3, fl, o);
        }
    INST_STAMP(7)
    if (A.stamps && threadIdx.x == 0) { A.stamps[5] += acc5; A.stamps[6] += acc6; A.stamps[7] += acc7; }
#undef INST_STAMP
}


// ---------------------------------------------------------------------------------------------------------------
// Class mode (stocs.cpp:363-519) in ONE launch: one 1024-thread workgroup per attempt (the attempts are independent) runs
// weights -> point 1 -> pass 1 -> compaction of the surviving weights -> points 2..4 with their passes over the survivors
// alone -> ordered base, with the attempt's weights in LDS.  The same stages as the instance kernel without the image-space
// ones.  A zero weight never changes a draw (the prefix sums are the same and the first index whose inclusive prefix
// exceeds r is a non-zero one), so drawing among the compacted survivors equals drawing among all points.
// Replaces nine launches (init, 4 x draw, 3 x pass, finalize) whose passes each walked every scene point of every attempt.
// ---------------------------------------------------------------------------------------------------------------
#define CLASS_TWO_LDS ((size_t)78 * 1024)     // two workgroups of class_attempts_kernel<true, true> on a CU: 2 x (this + ~1 KB of static LDS) <= 160 KB
struct ClassArgs {
    PassArgs pa;
    int draw_per_thread;        // (the prior every attempt starts from, stocs.cpp:372-381, is the .w of the scene positions)
    float* w_g; int32_t* sv_g;  // n_attempts x S each: the working set of scenes too large for LDS
    BaseOut* res;
    unsigned long long* stamps; // STOCS_DEBUG_TIMING only: cycles per stage of the first workgroup (else NULL)
    // trial batches (stocs_run_trials): workgroup a is attempt a % per_trial of trial a / per_trial and draws with that trial's seed
    const uint64_t* seeds;      // NULL: one trial, the kernel's seed argument
    int per_trial;
    int wg_offset;              // workgroup blockIdx.x is attempt slot wg_offset + blockIdx.x of the batch (launches of big scenes come in pieces)
    const int32_t* slot_list;   // != NULL: workgroup blockIdx.x redoes attempt slot slot_list[blockIdx.x] (the lean kernel's rare overflows)
};

// TWO: built for 64 VGPRs (a few spills) so that TWO workgroups share a CU when the attempt's LDS image allows it (scenes up to ~13 000
// points): the kernel waits on its bitmap probes and barriers most of the time, and a trial batch brings thousands of workgroups -- 64
// linemod trials 0.90 -> 0.65 ms of sampling (33 000 -> 36 600 trials/s; 1 024 trials in one call 45 900 -> 53 500).  The 80-VGPR build
// serves larger scenes (one workgroup per CU either way).
template <bool WLDS, bool TWO = false>
__global__ __launch_bounds__(1024, TWO ? 8 : 4) void class_attempts_kernel(ClassArgs A, uint64_t seed, int first_attempt, int n_attempts) {
    typedef typename InstTypes<WLDS>::sv_t sv_t;
    extern __shared__ __align__(16) unsigned char cls_dyn[];
    __shared__ uint64_t sh16[32];
    __shared__ int sh_pick[2];
    __shared__ int sh_ncand;
    __shared__ int sh_cnt[64], sh_cex[65];
    const int a = blockIdx.x;
    if (a >= n_attempts) return;
    const int S = A.pa.S;
    int slot = A.slot_list ? A.slot_list[a] : (A.seeds ? A.wg_offset + a : a);     // where the attempt's result goes
    int attempt = first_attempt + slot;
    if (A.seeds) { const int tr = slot / A.per_trial; seed = A.seeds[tr]; attempt = first_attempt + (slot - tr * A.per_trial); }
    float* w = WLDS ? (float*)cls_dyn : A.w_g + (size_t)a * S;
    sv_t* sv = WLDS ? (sv_t*)(cls_dyn + (((size_t)S * 4 + 15) & ~(size_t)15)) : (sv_t*)(A.sv_g + (size_t)a * S);
    const float4* spos = A.pa.spos;
    const float4* snrm = A.pa.snrm;
    BaseOut* out = A.res + slot;
    int32_t bidx[4] = {-1, -1, -1, -1};
    int fail = 0;
#define CLS_THREAD() int t = threadIdx.x; asm volatile("" : "+v"(t)); const int lane = t & 63, wv = t >> 6; (void)lane; (void)wv;
    unsigned long long tprev = A.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
#define CLS_STAMP(k) if (A.stamps && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); A.stamps[k] = now_ - tprev; tprev = now_; }
    // ---- "every base will start from the prior" (stocs.cpp:372-381) ----
    {
        CLS_THREAD()
        for (int i0 = 0; i0 < S; i0 += 8192) {                          // the class probability travels with the scene arrays (what the LCP adds, Q8)
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + k * 1024 + t; v[k] = i < S ? spos[i].w : 0.0f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + k * 1024 + t; if (i < S) w[i] = v[k]; }
        }
        if (t == 0) sh_ncand = 0;
    }
    __syncthreads();
    CLS_STAMP(0)
    bidx[0] = draw_block_fast(w, S, rng64(seed, (uint64_t)attempt, 0), sh16, sh_pick, 0, A.draw_per_thread);
    CLS_STAMP(1)
    if (bidx[0] < 0) fail = 1;   // "FAILED SAMPLING:: Zero probability returned" (:386-389); the same in every thread
    int n_surv = 0;
    if (!fail) {
        // ---- pass 1 (stocs.cpp:395-407), distance first: only the points that can have a key at all get their angles ----
        const int b1 = bidx[0];
        const float4 pc4 = spos[b1], nc4 = snrm[b1];
        const V3 pc = mk3(pc4.x, pc4.y, pc4.z), nc = mk3(nc4.x, nc4.y, nc4.z);
        {
            CLS_THREAD()
            for (int i0 = 0; i0 < S; i0 += 4096) {
                float wi[4]; float4 P[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * 1024 + t;
                    wi[k] = (i < S) ? w[i] : 0.0f;
                    P[k] = make_float4(0, 0, 0, 0);
                    if (wi[k] != 0.0f) P[k] = spos[i];
                }
                bool cand[4];
                unsigned long long cm[4];
                int n_here = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * 1024 + t;
                    cand[k] = false;
                    if (wi[k] != 0.0f) {
                        cand[k] = i != b1 && ppf_distance_may_have_key(A.pa.ix, pc - mk3(P[k].x, P[k].y, P[k].z));
                        if (!cand[k]) w[i] = 0.0f;
                    }
                    cm[k] = __ballot(cand[k]);
                    n_here += __popcll(cm[k]);
                }
                int base_pos = 0;
                if (lane == 0 && n_here) base_pos = atomicAdd(&sh_ncand, n_here);
                base_pos = __builtin_amdgcn_readfirstlane(base_pos);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (cand[k]) sv[base_pos + __popcll(cm[k] & ((1ull << lane) - 1ull))] = (sv_t)(i0 + k * 1024 + t);
                    base_pos += __popcll(cm[k]);
                }
            }
        }
        __syncthreads();
        CLS_STAMP(2)
        {
            CLS_THREAD()
            const int n_cand = sh_ncand;
            if (A.stamps && blockIdx.x == 0 && t == 0) A.stamps[8] = (unsigned long long)n_cand;
            for (int j0 = 0; j0 < n_cand; j0 += 2048) {
                int ii[2]; float4 P[2], N[2]; uint32_t key[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = j0 + k * 1024 + t;
                    ii[k] = -1; P[k] = make_float4(0, 0, 0, 0); N[k] = P[k];
                    if (j < n_cand) { ii[k] = (int)sv[j]; P[k] = spos[ii[k]]; N[k] = snrm[ii[k]]; }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    key[k] = PPF_NO_KEY;
                    if (ii[k] >= 0) key[k] = ppf_key_device(A.pa.ix, pc, nc, mk3(P[k].x, P[k].y, P[k].z), mk3(N[k].x, N[k].y, N[k].z));
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (ii[k] >= 0 && !ppf_key_present(A.pa.ix, key[k])) w[ii[k]] = 0.0f;
            }
        }
        __syncthreads();
        CLS_STAMP(3)
        // ---- the surviving weights compacted in scene order, in place (position <= index) ----
        {
            CLS_THREAD()
            for (int i0 = 0; i0 < S; i0 += 4096) {
                float wi[4];
                unsigned long long sbal[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * 1024 + t;
                    wi[k] = (i < S) ? w[i] : 0.0f;
                    sbal[k] = __ballot(wi[k] != 0.0f);
                    if (lane == 0) sh_cnt[k * 16 + wv] = __popcll(sbal[k]);
                }
                __syncthreads();
                if (wv == 0) {
                    const int v = sh_cnt[lane];
                    int inc = v;
                    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
                    sh_cex[lane] = inc - v;
                    if (lane == 63) sh_cex[64] = inc;
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (wi[k] != 0.0f) {
                        const int pos = n_surv + sh_cex[k * 16 + wv] + __popcll(sbal[k] & ((1ull << lane) - 1ull));
                        sv[pos] = (sv_t)(i0 + k * 1024 + t); w[pos] = wi[k];
                    }
                n_surv += sh_cex[64];
                __syncthreads();
            }
        }
        CLS_STAMP(4)
        if (A.stamps && blockIdx.x == 0 && threadIdx.x == 0) A.stamps[9] = (unsigned long long)n_surv;
        // ---- points 2..4 (stocs.cpp:410-505) over the survivors ----
        for (int k = 1; k < 4 && !fail; ++k) {
            CLS_THREAD()
            const int pos = draw_block_fast(w, n_surv, rng64(seed, (uint64_t)attempt, (uint64_t)k), sh16, sh_pick, k & 1, A.draw_per_thread);
            if (pos < 0) { fail = 1; break; }
            bidx[k] = (int32_t)sv[pos];
            if (k < 3) {
                for (int j = t; j < n_surv; j += 1024) {
                    if (w[j] == 0.0f) continue;                               // already zero: nothing to decide
                    const int i = (int)sv[j];
                    const bool z = (k == 1) ? pass_zeroes<2>(A.pa, bidx[0], bidx[1], -1, i) : pass_zeroes<3>(A.pa, bidx[0], bidx[1], bidx[2], i);
                    if (z) w[j] = 0.0f;
                }
                __syncthreads();
            }
        }
    }
#undef CLS_THREAD
    CLS_STAMP(5)
    if (threadIdx.x < 64) finalize_one_wave_call(A.pa.spos, bidx[0], bidx[1], bidx[2], bidx[3], fail, out);
    CLS_STAMP(6)
#undef CLS_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// The lean form of class_attempts_kernel (round 5): the same attempt, the same bases bit for bit, in 2 bytes of LDS per scene point
// instead of 6 -- so that TWO 1024-thread workgroups share a CU on the ycb frame (14 247 points: 85 KB -> 32 KB) and on the metric
// scene (20 000: 120 KB -> 44 KB), where a small-frame trial batch spends most of its time (64 ycb trials: 6 400 attempts on 256 CUs).
//   * every attempt starts from the prior (stocs.cpp:372-381), so point 1 is drawn from ONE table for all attempts: the inclusive prefix
//     sums of the 2^32 fixed-point prior weights in scene order (prior_cdf_kernel + scan, once per prior); the draw of
//     sample_point_from_distribution (stocs.cpp:133-148; seeded, Q6) -- first index whose inclusive prefix exceeds mulhi64(r, total) -- is a
//     64-ary search by one wavefront: three dependent loads instead of a scan over the scene.  No copy of the prior in LDS;
//   * pass 1 keeps a LIST of the points within key distance of point 1 (u16 indices) and marks the survivors of its angle test in a bitmap;
//   * the survivors -- a third of the scene at the very most, else the attempt is flagged and redone by class_attempts_kernel -- are
//     compacted from the bitmap in scene order over the dead list: index + weight, 6 bytes each; points 2-4 run on them as before.
// ---------------------------------------------------------------------------------------------------------------
#define LEAN_MAX_S 32768
#define LEAN_QUARTER_S 8000
#define LEAN_HALF_S 24000   // scenes up to here run the lean kernel with 512 threads, four workgroups per CU (see lean_lds_bytes)
__global__ __launch_bounds__(256) void prior_fix_kernel(const float4* __restrict__ spos, int S, unsigned long long* __restrict__ fix) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= S) fix[i] = i < S ? weight_fix_dev(spos[i].w) : 0ull;      // (S + 1 entries: the exclusive scan's last one is the total)
}

template <int NT>
__global__ __launch_bounds__(NT, 8) void class_attempts_lean_kernel(ClassArgs A, uint64_t seed, int first_attempt, int n_attempts,
                                                                      const unsigned long long* __restrict__ cdf_excl, int cap) {
    extern __shared__ __align__(16) unsigned char lean_dyn[];       // the candidate list of pass 1 (u16 x S), then the survivors (f32 + u16) x cap
    __shared__ uint32_t sh_alive[NT == 256 ? (LEAN_QUARTER_S + 31) / 32 + 1 : (NT == 512 ? (LEAN_HALF_S + 31) / 32 + 1 : LEAN_MAX_S / 32)];   // (the small form: 2.3 KB -- with 36 KB of dynamic LDS four workgroups fit a CU)
    __shared__ uint64_t sh16[32];
    __shared__ int sh_pick[2];
    __shared__ int sh_ncand, sh_b1;
    __shared__ int sh_cnt[NT / 64];
    const int a = blockIdx.x;
    if (a >= n_attempts) return;
    const int S = A.pa.S;
    const int slot = A.seeds ? A.wg_offset + a : a;
    int attempt = first_attempt + slot;
    if (A.seeds) { const int tr = slot / A.per_trial; seed = A.seeds[tr]; attempt = first_attempt + (slot - tr * A.per_trial); }
    uint16_t* cl = (uint16_t*)lean_dyn;                              // candidates of pass 1
    float* w = (float*)lean_dyn;                                     // survivors' weights ...
    uint16_t* sv = (uint16_t*)(lean_dyn + (size_t)cap * 4);          // ... and indices (behind the weights)
    const float4* spos = A.pa.spos;
    const float4* snrm = A.pa.snrm;
    const unsigned long long* cdf = cdf_excl + 1;                    // inclusive prefix of point i
    BaseOut* out = A.res + slot;
    int32_t bidx[4] = {-1, -1, -1, -1};
    int fail = 0;
#define CLS_THREAD() int t = threadIdx.x; asm volatile("" : "+v"(t)); const int lane = t & 63, wv = t >> 6; (void)lane; (void)wv;
    // ---- point 1: 64-ary search of the prior's prefix sums by the first wavefront ----
    {
        CLS_THREAD()
        if (t == 0) sh_ncand = 0;
        for (int i = t; i < (S + 31) / 32; i += NT) sh_alive[i] = 0u;
        if (wv == 0) {
            const unsigned long long total = cdf[S - 1];
            int pick = -1;
            if (total != 0ull) {
                const unsigned long long r = mulhi64(rng64(seed, (uint64_t)attempt, 0), total);
                int lo = 0, hi = S - 1;                              // cdf[hi] > r; the answer -- the first index whose prefix exceeds r -- lies in [lo, hi]
                while (lo < hi) {
                    const int len = hi - lo + 1, step = (len + 63) >> 6;
                    const int pidx = lo + lane * step;
                    const bool f = pidx <= hi ? (cdf[pidx] > r) : true;
                    const unsigned long long m = __ballot(f);
                    if (m == 0ull) { lo = lo + 63 * step + 1; continue; }     // (every probe below hi and none exceeds r: the answer is behind the last probe)
                    const int k = (int)__builtin_ctzll(m);
                    hi = min(lo + k * step, hi);
                    lo = k ? lo + (k - 1) * step + 1 : lo;
                    if (k == 0) hi = lo;
                }
                pick = lo;
            }
            if (lane == 0) sh_b1 = pick;
        }
    }
    __syncthreads();
    bidx[0] = sh_b1;
    if (bidx[0] < 0) fail = 1;   // "FAILED SAMPLING:: Zero probability returned" (:386-389)
    int n_surv = 0;
    if (!fail) {
        // ---- pass 1 (stocs.cpp:395-407), distance first: the points that can have a key at all go to the list ----
        const int b1 = bidx[0];
        const float4 pc4 = spos[b1], nc4 = snrm[b1];
        const V3 pc = mk3(pc4.x, pc4.y, pc4.z), nc = mk3(nc4.x, nc4.y, nc4.z);
        {
            CLS_THREAD()
            for (int i0 = 0; i0 < S; i0 += 4 * NT) {
                float4 P[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int i = i0 + k * NT + t; P[k] = i < S ? spos[i] : make_float4(0, 0, 0, 0); }
                bool cand[4];
                unsigned long long cm[4];
                int n_here = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + k * NT + t;
                    cand[k] = i < S && P[k].w != 0.0f && i != b1 && ppf_distance_may_have_key(A.pa.ix, pc - mk3(P[k].x, P[k].y, P[k].z));
                    cm[k] = __ballot(cand[k]);
                    n_here += __popcll(cm[k]);
                }
                int base_pos = 0;
                if (lane == 0 && n_here) base_pos = atomicAdd(&sh_ncand, n_here);
                base_pos = __builtin_amdgcn_readfirstlane(base_pos);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (cand[k]) cl[base_pos + __popcll(cm[k] & ((1ull << lane) - 1ull))] = (uint16_t)(i0 + k * NT + t);
                    base_pos += __popcll(cm[k]);
                }
            }
        }
        __syncthreads();
        {   // the angles of the listed points: a point whose key the model has survives (one bit)
            CLS_THREAD()
            const int n_cand = sh_ncand;
            for (int j0 = 0; j0 < n_cand; j0 += 2 * NT) {
                int ii[2]; float4 P[2], N[2]; uint32_t key[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = j0 + k * NT + t;
                    ii[k] = -1; P[k] = make_float4(0, 0, 0, 0); N[k] = P[k];
                    if (j < n_cand) { ii[k] = (int)cl[j]; P[k] = spos[ii[k]]; N[k] = snrm[ii[k]]; }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    key[k] = PPF_NO_KEY;
                    if (ii[k] >= 0) key[k] = ppf_key_device(A.pa.ix, pc, nc, mk3(P[k].x, P[k].y, P[k].z), mk3(N[k].x, N[k].y, N[k].z));
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (ii[k] >= 0 && ppf_key_present(A.pa.ix, key[k])) atomicOr(&sh_alive[ii[k] >> 5], 1u << (ii[k] & 31));
            }
        }
        __syncthreads();
        {   // ---- the survivors in scene order: thread t owns 32 * WPT consecutive points (WPT words of the bitmap) ----
            CLS_THREAD()
            constexpr int WPT = LEAN_MAX_S / 32 / NT;                 // 1 with 1024 threads, 2 with 512
            uint32_t bw[WPT];
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < WPT; ++q) { const int wi = t * WPT + q; bw[q] = wi < (S + 31) / 32 ? sh_alive[wi] : 0u; cnt += __popc(bw[q]); }
            int inc = cnt;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
            if (lane == 63) sh_cnt[wv] = inc;
            __syncthreads();
            int pre = 0, tot = 0;
            for (int x = 0; x < NT / 64; ++x) { const int v = sh_cnt[x]; if (x < wv) pre += v; tot += v; }
            n_surv = tot;
            if (n_surv > cap) {                                   // (uniform) more survivors than the list holds: the full-size kernel redoes this attempt
                if (t == 0) { for (int k = 0; k < 4; ++k) out->ids[k] = -1; out->inv[0] = out->inv[1] = 0.0f; out->valid = 0; out->pad = 1; }
                return;
            }
            int pos = pre + inc - cnt;
#pragma unroll
            for (int q = 0; q < WPT; ++q) {
                uint32_t b = bw[q];
                while (b) {
                    const int i = 32 * (t * WPT + q) + (int)__builtin_ctz(b);
                    b &= b - 1u;
                    sv[pos] = (uint16_t)i; w[pos] = spos[i].w;   // (the candidate list is dead: every thread is behind the barrier above)
                    ++pos;
                }
            }
        }
        __syncthreads();
        // ---- points 2..4 (stocs.cpp:410-505) over the survivors ----
        for (int k = 1; k < 4 && !fail; ++k) {
            CLS_THREAD()
            const int pos = draw_block_fast(w, n_surv, rng64(seed, (uint64_t)attempt, (uint64_t)k), sh16, sh_pick, k & 1, A.draw_per_thread);
            if (pos < 0) { fail = 1; break; }
            bidx[k] = (int32_t)sv[pos];
            if (k < 3) {
                for (int j = t; j < n_surv; j += NT) {
                    if (w[j] == 0.0f) continue;                               // already zero: nothing to decide
                    const int i = (int)sv[j];
                    const bool z = (k == 1) ? pass_zeroes<2>(A.pa, bidx[0], bidx[1], -1, i) : pass_zeroes<3>(A.pa, bidx[0], bidx[1], bidx[2], i);
                    if (z) w[j] = 0.0f;
                }
                __syncthreads();
            }
        }
    }
#undef CLS_THREAD
    if (threadIdx.x < 64) finalize_one_wave_call(A.pa.spos, bidx[0], bidx[1], bidx[2], bidx[3], fail, out);
}

// the prior's prefix sums for the lean kernel: (re)computed when the class probabilities on the device have changed since the last time
static int ensure_prior_cdf(stocs_ctx* c) {
    const size_t S = (size_t)c->nS;
    if (c->d_cdf && c->cdf_epoch == c->prior_epoch && c->cdf_n == S) return STOCS_OK;
    size_t tb = 0;
    STOCS_HIP_CHECK(exclusive_scan(NULL, tb, (const unsigned long long*)NULL, (unsigned long long*)NULL, S + 1, c->stream));
    const size_t need = 2 * ((S + 1) * 8 + 256) + tb + 256;
    if (c->cdf_bytes < need) {
        if (c->d_cdf) { STOCS_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_cdf); c->d_cdf = NULL; c->cdf_bytes = 0; }
        STOCS_HIP_CHECK(dev_malloc((void**)&c->d_cdf, need + need / 4));
        c->cdf_bytes = need + need / 4;
    }
    unsigned long long* cdf = (unsigned long long*)c->d_cdf;
    unsigned long long* fix = (unsigned long long*)((char*)c->d_cdf + (((S + 1) * 8 + 255) & ~(size_t)255));
    void* tmp = (char*)fix + (((S + 1) * 8 + 255) & ~(size_t)255);
    hipLaunchKernelGGL(prior_fix_kernel, dim3((unsigned)((S + 256) / 256)), dim3(256), 0, c->stream, (const float4*)c->d_spos, (int)S, fix);
    STOCS_HIP_CHECK(exclusive_scan(tmp, tb, (const unsigned long long*)fix, cdf, S + 1, c->stream));
    c->cdf_epoch = c->prior_epoch; c->cdf_n = S;
    return STOCS_OK;
}

// LDS of the lean kernel and the survivors it holds: at least 2 bytes per scene point (the candidate list), reused as 6 bytes per survivor;
// up to 75 KB -- what still lets two workgroups share a CU -- are taken, so that small scenes hold every point (no attempt is ever redone) and
// the ycb frame / the metric scene 12 800 survivors (an attempt there keeps 1 600-4 000)
// Round 5b: scenes of at most LEAN_HALF_S points run the kernel with 512 threads and at most 36 KB of dynamic LDS per workgroup (40 KB for the
// metric scene's 20 000 points) -- FOUR attempts per CU instead of two (three at the metric size): the kernel is a chain of ~15 barrier-separated
// stages with no unit above 0.42 busy, and twice as many independent chains per CU fill the gaps (each attempt takes longer, the CU finishes
// more of them): 64 ycb trials' sampling 0.78 -> 0.63 ms, linemod 0.66 -> 0.51, Cm 1.10 -> 1.03.  Scenes of at most LEAN_QUARTER_S points go one step
// further: 256 threads, 16 KB, eight attempts per CU (linemod 0.51 -> 0.46 ms).  STOCS_CLASS_LEAN_1024 / STOCS_CLASS_LEAN_512 keep the larger forms.
static inline bool lean_half(size_t S) { return S <= LEAN_HALF_S && !getenv("STOCS_CLASS_LEAN_1024"); }
static inline bool lean_quarter(size_t S) { return S <= LEAN_QUARTER_S && lean_half(S) && !getenv("STOCS_CLASS_LEAN_512"); }   // 256 threads, eight attempts per CU
static inline size_t lean_lds_bytes(size_t S) {
    const size_t top = lean_quarter(S) ? 16384 : (lean_half(S) ? 36864 : 76800);
    return std::max((S * 2 + 15) & ~(size_t)15, std::min<size_t>(top, (S * 6 + 31) & ~(size_t)15));
}
static inline int lean_cap(size_t S) {
    int cap = (int)std::min<size_t>(S + 1, (lean_lds_bytes(S) - 8) / 6) & ~1;
    if (const char* e = getenv("STOCS_CLASS_LEAN_CAP")) cap = std::max(2, std::min(cap, atoi(e) & ~1));   // (tests: forces the overflow path)
    return cap;
}
static inline bool lean_usable(const stocs_ctx* c) { return c->nS >= 64 && c->nS <= LEAN_MAX_S && !getenv("STOCS_CLASS_FULL_KERNEL") && !getenv("STOCS_INSTANCE_NO_LDS"); }

// attempts the lean kernel flagged (more survivors than its list holds; pad == 1 in their result): redone by the full-size kernel, in place
static int redo_lean_overflows(stocs_ctx* c, ClassArgs A, uint64_t seed, int first_attempt, BaseOut* res_host, size_t n, int32_t* d_slots) {
    std::vector<int32_t> slots;
    for (size_t i = 0; i < n; ++i) if (res_host[i].pad == 1) slots.push_back((int32_t)i);
    if (slots.empty()) return STOCS_OK;
    const size_t S = (size_t)c->nS;
    int rc = STOCS_OK;
    do {
        if (hipMemcpyAsync(d_slots, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = STOCS_ERR_HIP; break; }
        A.slot_list = d_slots;
        const size_t lds = ((S * 4 + 15) & ~(size_t)15) + S * 2 + 16;
        if (hipFuncSetAttribute((const void*)class_attempts_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) != hipSuccess) { rc = STOCS_ERR_HIP; break; }
        hipLaunchKernelGGL(class_attempts_kernel<true>, dim3((unsigned)slots.size()), dim3(1024), lds, c->stream, A, seed, first_attempt, (int)slots.size());
        if (hipGetLastError() != hipSuccess) { rc = STOCS_ERR_HIP; break; }
        for (size_t k = 0; k < slots.size() && !rc; ++k)
            if (hipMemcpyAsync(&res_host[slots[k]], A.res + slots[k], sizeof(BaseOut), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = STOCS_ERR_HIP;
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = STOCS_ERR_HIP;
    } while (0);
    if (rc) set_error("class-mode sampling: redoing %zu attempts with the full-size kernel failed", slots.size());
    return rc;
}

// class mode through the one-launch kernel; scenes beyond the LDS working set keep it in device memory (same code)
static int sample_class(stocs_ctx* c, uint64_t seed, int first_attempt, int nB, int32_t* ids, float* inv, int32_t* valid) {
    if (getenv("STOCS_CLASS_MULTI_KERNEL")) return sample_class_multi(c, seed, first_attempt, nB, ids, inv, valid);   // the nine-launch form (A/B)
    const size_t S = (size_t)c->nS;
    const bool wlds = S <= 26000 && !getenv("STOCS_INSTANCE_NO_LDS");        // 6 bytes per point of the 160 KB
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t b_res = al((size_t)nB * sizeof(BaseOut)), b_w = wlds ? 0 : al((size_t)nB * S * 4), b_sv = wlds ? 0 : al((size_t)nB * S * 4), b_slots = al((size_t)nB * 4);
    int rc = ensure_scratch(c, b_res + b_w + b_sv + b_slots + 256);
    if (rc) return rc;
    char* p = (char*)c->d_scratch;
    ClassArgs A;
    A.pa = pass_args(c);
    A.res = (BaseOut*)p; p += b_res;
    A.w_g = (float*)p; p += b_w;
    A.sv_g = (int32_t*)p;
    int32_t* d_slots = (int32_t*)((char*)c->d_scratch + b_res + b_w + b_sv + 256);   // (behind the debug stamps)
    A.draw_per_thread = 2;
    A.seeds = NULL; A.per_trial = 0; A.wg_offset = 0; A.slot_list = NULL;
    const bool dbg = getenv("STOCS_DEBUG_TIMING") != NULL;
    A.stamps = NULL;
    if (dbg) {   // behind everything else in the scratch area
        rc = ensure_scratch(c, b_res + b_w + b_sv + 256);
        if (rc) return rc;
        p = (char*)c->d_scratch;
        A.res = (BaseOut*)p; A.w_g = (float*)(p + b_res); A.sv_g = (int32_t*)(p + b_res + b_w);
        A.stamps = (unsigned long long*)(p + b_res + b_w + b_sv);
        STOCS_HIP_CHECK(hipMemsetAsync(A.stamps, 0, 128, c->stream));
    }
    const size_t lds = wlds ? ((S * 4 + 15) & ~(size_t)15) + S * 2 + 16 : 0;
    // (a single trial's 100 attempts have a CU each either way: the lean kernel pays when there are more workgroups than CUs, or when the
    //  prior's prefix sums exist already -- a new frame would otherwise pay 13 us for three small launches it has no use for)
    const bool lean = wlds && lean_usable(c) && !dbg && (nB > 256 || (c->d_cdf && c->cdf_epoch == c->prior_epoch && c->cdf_n == S) || getenv("STOCS_CLASS_LEAN_KERNEL"));
    if (lean) {      // 2 bytes of LDS per scene point: two workgroups per CU (the rare attempt with too many survivors is redone below)
        if ((rc = ensure_prior_cdf(c))) return rc;
        if (lean_quarter(S)) hipLaunchKernelGGL(class_attempts_lean_kernel<256>, dim3((unsigned)nB), dim3(256), lean_lds_bytes(S), c->stream, A, seed, first_attempt, nB, (const unsigned long long*)c->d_cdf, lean_cap(S));
        else if (lean_half(S)) hipLaunchKernelGGL(class_attempts_lean_kernel<512>, dim3((unsigned)nB), dim3(512), lean_lds_bytes(S), c->stream, A, seed, first_attempt, nB, (const unsigned long long*)c->d_cdf, lean_cap(S));
        else hipLaunchKernelGGL(class_attempts_lean_kernel<1024>, dim3((unsigned)nB), dim3(1024), lean_lds_bytes(S), c->stream, A, seed, first_attempt, nB, (const unsigned long long*)c->d_cdf, lean_cap(S));
    } else
    if (wlds && lds <= CLASS_TWO_LDS && nB > 256) {     // (more workgroups than CUs: two per CU pay)
        STOCS_HIP_CHECK(hipFuncSetAttribute((const void*)class_attempts_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLASS_TWO_LDS));
        hipLaunchKernelGGL((class_attempts_kernel<true, true>), dim3((unsigned)nB), dim3(1024), lds, c->stream, A, seed, first_attempt, nB);
    } else if (wlds) {
        STOCS_HIP_CHECK(hipFuncSetAttribute((const void*)class_attempts_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        hipLaunchKernelGGL(class_attempts_kernel<true>, dim3((unsigned)nB), dim3(1024), lds, c->stream, A, seed, first_attempt, nB);
    } else {
        hipLaunchKernelGGL(class_attempts_kernel<false>, dim3((unsigned)nB), dim3(1024), 0, c->stream, A, seed, first_attempt, nB);
    }
    STOCS_HIP_CHECK(hipGetLastError());
    // (the attempts' results come back through the context's pinned block: a copy into a pageable std::vector takes the runtime's staging path,
    //  ~30 us of a trial)
    if ((rc = ensure_pinned(c, (size_t)PIN_VAR + (size_t)nB * sizeof(BaseOut) + 256))) return rc;
    BaseOut* res_pin = (BaseOut*)((char*)c->h_pin + PIN_VAR);
    STOCS_HIP_CHECK(hipMemcpyAsync(res_pin, A.res, (size_t)nB * sizeof(BaseOut), hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<BaseOut> res(res_pin, res_pin + nB);
    if (lean && (rc = redo_lean_overflows(c, A, seed, first_attempt, res.data(), (size_t)nB, d_slots))) return rc;
    if (dbg) {
        unsigned long long st[16];
        STOCS_HIP_CHECK(hipMemcpy(st, A.stamps, 128, hipMemcpyDeviceToHost));
        fprintf(stderr, "[stocs class] first attempt, shader cycles: prior %llu | draw 1 %llu | pass 1 distances %llu (%llu of %d in range) | pass 1 angles %llu | compaction %llu (%llu survivors) | points 2-4 %llu | ordered base %llu\n",
                st[0], st[1], st[2], st[8], c->nS, st[3], st[4], st[9], st[5], st[6]);
    }
    return record_bases(c, nB, res.data(), ids, inv, valid);
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
    c->prior_epoch++;       // (the prefix sums the lean class kernel draws point 1 from are stale)
    return STOCS_OK;
}

// Device-side state of instance mode, owned by the context (stocs_ctx::inst)
struct InstanceState {
    bool runs_valid = false;     // runs / pt_run / edge_pt match the current edge map and scene
    int S = 0, Sw = 0;
    char* d_mem = NULL; size_t mem_bytes = 0;
    uint16_t* d_run_s = NULL; uint16_t* d_run_e = NULL; uint32_t* d_row_off = NULL;
    int32_t* d_pt_run = NULL; uint8_t* d_edge_pt = NULL; uint8_t* d_prev_in = NULL; uint8_t* d_label = NULL;
    float* d_cls = NULL; uint32_t* d_maskbits = NULL; uint32_t* d_segbits = NULL; uint32_t* d_parent = NULL;
    int32_t* d_sv = NULL; float* d_w = NULL;
    RunPair* d_pairs = NULL; uint32_t* d_pair_off = NULL;
    char* d_queue = NULL; size_t queue_bytes = 0;   // hand-over slots between the two workgroups (grown on demand)
    char* d_trials = NULL; size_t trials_bytes = 0; // per-trial copies of the mutable state of a trial batch (stocs_run_trials)
    size_t n_runs = 0;
    std::vector<uint32_t> h_segbits;
};

static void free_instance_state(stocs_ctx* c) {
    InstanceState* I = (InstanceState*)c->inst;
    if (!I) return;
    if (I->d_mem) (void)hipFree(I->d_mem);
    if (I->d_queue) (void)hipFree(I->d_queue);
    if (I->d_trials) (void)hipFree(I->d_trials);
    delete I;
    c->inst = NULL;
}

// (re)builds what depends on the edge map and the scene's pixels; clears the per-trial state
static int prepare_instance_state(stocs_ctx* c) {
    if (!c->inst) c->inst = new InstanceState();
    InstanceState* I = (InstanceState*)c->inst;
    if (I->runs_valid && I->S == c->nS) return STOCS_OK;
    const int S = c->nS, W = c->prm.image_width, H = c->prm.image_height;
    // passable runs (png value 255 <=> edge_probability == 0, rgbd.cpp:350) row by row
    std::vector<uint16_t> rs, re;
    std::vector<uint32_t> row_off((size_t)H + 1, 0);
    for (int r = 0; r < H; ++r) {
        row_off[r] = (uint32_t)rs.size();
        const uint8_t* row = &c->edge_map[(size_t)r * W];
        int col = 0;
        while (col < W) {
            if (row[col] != 255) { ++col; continue; }
            const int s0 = col;
            while (col < W && row[col] == 255) ++col;
            rs.push_back((uint16_t)s0); re.push_back((uint16_t)col);
        }
    }
    row_off[H] = (uint32_t)rs.size();
    // the run graph: runs of consecutive rows whose pixels touch in the 8-neighbourhood (columns [s, e): s_h <= e_g and e_h >= s_g)
    std::vector<RunPair> pairs;
    std::vector<uint32_t> pair_off((size_t)H + 1, 0);
    for (int r = 0; r < H; ++r) {
        pair_off[r] = (uint32_t)pairs.size();
        if (r + 1 >= H) continue;
        const uint32_t a0 = row_off[r], a1 = row_off[r + 1], b1 = row_off[r + 2];
        uint32_t h0 = a1;
        for (uint32_t g = a0; g < a1; ++g) {
            while (h0 < b1 && re[h0] < rs[g]) ++h0;
            for (uint32_t h = h0; h < b1 && rs[h] <= re[g]; ++h) {
                RunPair q;
                q.g = g; q.dh = (uint16_t)(h - g); q.row = (uint16_t)r; q.gs = rs[g]; q.ge = re[g]; q.hs = rs[h]; q.he = re[h];
                pairs.push_back(q);
            }
        }
    }
    pair_off[H] = (uint32_t)pairs.size();
    std::vector<int32_t> pt_run(S, -1);
    std::vector<uint8_t> edge_pt(S, 0);
    for (int i = 0; i < S; ++i) {
        const int r = c->h_spix[2 * i], col = c->h_spix[2 * i + 1];
        const uint8_t v = c->edge_map[(size_t)r * W + col];
        edge_pt[i] = ((float)(255.0 - v) / 255.0 == 1) ? 1 : 0;    // prune_edge_pixels, stocs.cpp:529-533
        if (v == 255) {   // binary search of the run that holds the column
            uint32_t lo = row_off[r], hi = row_off[r + 1];
            while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (rs[mid] <= col) lo = mid; else hi = mid; }
            pt_run[i] = (int32_t)lo;
        }
    }
    const int Sw = ((S + 63) / 64) * 2;   // whole 64-bit ballots
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nr = std::max<size_t>(rs.size(), 1);
    const size_t o_rs = 0, o_re = o_rs + al(nr * 2), o_ro = o_re + al(nr * 2), o_pr = o_ro + al(((size_t)H + 1) * 4), o_ep = o_pr + al((size_t)S * 4),
                 o_pi = o_ep + al(S), o_lb = o_pi + al(S), o_cl = o_lb + al(S), o_mb = o_cl + al((size_t)S * 4), o_sb = o_mb + al((size_t)256 * Sw * 4),
                 o_pa = o_sb + al((size_t)Sw * 4), o_sv = o_pa + al((nr + 1) * 4), o_wc = o_sv + al((size_t)S * 4), o_pp = o_wc + al((size_t)S * 4),
                 o_po = o_pp + al(std::max<size_t>(pairs.size(), 1) * sizeof(RunPair)), total = o_po + al(((size_t)H + 1) * 4);
    if (I->mem_bytes < total) {
        if (I->d_mem) { STOCS_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipFree(I->d_mem); I->d_mem = NULL; I->mem_bytes = 0; }
        STOCS_HIP_CHECK(dev_malloc((void**)&I->d_mem, total + total / 4));
        I->mem_bytes = total + total / 4;
    }
    char* m = I->d_mem;
    I->d_run_s = (uint16_t*)(m + o_rs); I->d_run_e = (uint16_t*)(m + o_re); I->d_row_off = (uint32_t*)(m + o_ro); I->d_pt_run = (int32_t*)(m + o_pr);
    I->d_edge_pt = (uint8_t*)(m + o_ep); I->d_prev_in = (uint8_t*)(m + o_pi); I->d_label = (uint8_t*)(m + o_lb); I->d_cls = (float*)(m + o_cl);
    I->d_maskbits = (uint32_t*)(m + o_mb); I->d_segbits = (uint32_t*)(m + o_sb); I->d_parent = (uint32_t*)(m + o_pa);
    I->d_sv = (int32_t*)(m + o_sv); I->d_w = (float*)(m + o_wc);
    I->d_pairs = (RunPair*)(m + o_pp); I->d_pair_off = (uint32_t*)(m + o_po);
    I->S = S; I->Sw = Sw; I->n_runs = rs.size();
    hipStream_t st = c->stream;
    if (!rs.empty()) {
        STOCS_HIP_CHECK(hipMemcpyAsync(I->d_run_s, rs.data(), rs.size() * 2, hipMemcpyHostToDevice, st));
        STOCS_HIP_CHECK(hipMemcpyAsync(I->d_run_e, re.data(), re.size() * 2, hipMemcpyHostToDevice, st));
    }
    if (!pairs.empty()) STOCS_HIP_CHECK(hipMemcpyAsync(I->d_pairs, pairs.data(), pairs.size() * sizeof(RunPair), hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(I->d_pair_off, pair_off.data(), ((size_t)H + 1) * 4, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(I->d_row_off, row_off.data(), ((size_t)H + 1) * 4, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(I->d_pt_run, pt_run.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(I->d_edge_pt, edge_pt.data(), S, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemcpyAsync(I->d_cls, c->h_sprob.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_prev_in, 0, S, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_label, 0, S, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_maskbits, 0, (size_t)256 * Sw * 4, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_segbits, 0, (size_t)Sw * 4, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));   // the host vectors above die here
    I->runs_valid = true;
    return STOCS_OK;
}

// a new trial stream on the same scene and edge map: prior restored, image-space state cleared
static int reset_instance_trial(stocs_ctx* c) {
    InstanceState* I = (InstanceState*)c->inst;
    if (!I || !I->runs_valid || I->S != c->nS) return STOCS_OK;
    hipStream_t st = c->stream;
    STOCS_HIP_CHECK(hipMemcpyAsync(I->d_cls, c->h_sprob.data(), (size_t)I->S * 4, hipMemcpyHostToDevice, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_prev_in, 0, I->S, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_label, 0, I->S, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_maskbits, 0, (size_t)256 * I->Sw * 4, st));
    STOCS_HIP_CHECK(hipMemsetAsync(I->d_segbits, 0, (size_t)I->Sw * 4, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    I->h_segbits.clear();
    return STOCS_OK;
}

static int sample_instance(stocs_ctx* c, uint64_t seed, int first_attempt, int nB, float dispersion, int32_t* ids, float* inv, int32_t* valid) {
    int rc = prepare_instance_state(c);
    if (rc) return rc;
    InstanceState* I = (InstanceState*)c->inst;
    SampleBuffers sb;
    if ((rc = carve(c, std::max(nB, 1), &sb))) return rc;
    InstanceArgs A;
    A.pa = pass_args(c);
    A.pix = c->d_spix; A.edge_pt = I->d_edge_pt; A.pt_run = I->d_pt_run; A.run_s = I->d_run_s; A.run_e = I->d_run_e; A.row_off = I->d_row_off;
    A.pairs = I->d_pairs; A.pair_off = I->d_pair_off;
    A.draw_per_thread = 2;   // measured on the packed frame (3 415 points, ~290 survivors): 1..4 within 2 %, 8 and 16 slower
    if (const char* e = getenv("STOCS_DRAW_PER_THREAD")) A.draw_per_thread = std::max(1, atoi(e));
    A.H = c->prm.image_height; A.W = c->prm.image_width; A.Sw = I->Sw;
    A.cls = I->d_cls; A.prev_in = I->d_prev_in; A.label = I->d_label; A.maskbits = I->d_maskbits; A.segbits = I->d_segbits; A.parent_g = I->d_parent;
    const bool dbg = getenv("STOCS_DEBUG_TIMING") != NULL;
    A.stamps = NULL;
    if (dbg) { A.stamps = (unsigned long long*)I->d_parent; STOCS_HIP_CHECK(hipMemsetAsync(I->d_parent, 0, 128, c->stream)); }   // parent_g is idle for small discs
    A.w = I->d_w; A.sv = I->d_sv; A.spos_w = c->d_spos; A.snrm_w = c->d_snrmw; A.res = sb.res;
    c->prior_epoch++;       // (the kernel writes the decayed prior into the scene arrays)
    A.n_trials = 0; A.trial_stride = 0; A.seeds = NULL;
    // hand-over slots of the attempts: header, S (index, weight) pairs, flag; one error word
    {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t nq = (size_t)std::max(nB, 1), Sq = (size_t)c->nS;
        const size_t o_hdr = 0, o_flag = o_hdr + al(nq * 16), o_err = o_flag + al(nq * 4), o_sv = o_err + 256, o_w = o_sv + al(nq * Sq * 4), total = o_w + al(nq * Sq * 4);
        if (I->queue_bytes < total) {
            if (I->d_queue) { STOCS_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipFree(I->d_queue); I->d_queue = NULL; I->queue_bytes = 0; }
            // sized for a whole trial of this scene at once (<= 254 attempts), so that a caller asking attempt by attempt allocates once
            const size_t full = o_w - o_sv > 0 ? 256 + al(254 * 16) + al(254 * 4) + 2 * al((size_t)254 * Sq * 4) : total;
            const size_t want = std::max(total, std::min<size_t>(full, (size_t)1 << 30));
            STOCS_HIP_CHECK(dev_malloc((void**)&I->d_queue, want));
            I->queue_bytes = want;
        }
        A.q_hdr = (int4*)(I->d_queue + o_hdr); A.q_flag = (unsigned int*)(I->d_queue + o_flag); A.q_err = (unsigned int*)(I->d_queue + o_err);
        A.q_sv = (int32_t*)(I->d_queue + o_sv); A.q_w = (float*)(I->d_queue + o_w);
        STOCS_HIP_CHECK(hipMemsetAsync(I->d_queue + o_flag, 0, o_sv - o_flag, c->stream));   // flags and the error word
    }
    // parents, and up to INST_LDS_POINTS points' weights + survivor indices, in LDS (<= 160 KB per workgroup on gfx950)
    const size_t lds_parent = ((size_t)(INST_MAX_NODES + 1) * 4 + 15) & ~(size_t)15;
    const bool wlds = c->nS <= INST_LDS_POINTS && !getenv("STOCS_INSTANCE_NO_LDS");
    const size_t lds = lds_parent + (wlds ? ((((size_t)c->nS * 4 + 15) & ~(size_t)15) + (size_t)c->nS * 2 + 16) : 0);
    {
        const void* fn = wlds ? (const void*)instance_attempts_kernel<true> : (const void*)instance_attempts_kernel<false>;
        STOCS_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    }
    unsigned n_wg = 2;
    if (const char* e = getenv("STOCS_INSTANCE_GRID")) n_wg = (unsigned)std::max(2, atoi(e));   // 9: both working workgroups on one XCD (A/B of the placement)
    if (wlds) hipLaunchKernelGGL(instance_attempts_kernel<true>, dim3(n_wg), dim3(1024), lds, c->stream, A, seed, first_attempt, nB, dispersion);
    else hipLaunchKernelGGL(instance_attempts_kernel<false>, dim3(n_wg), dim3(1024), lds, c->stream, A, seed, first_attempt, nB, dispersion);
    STOCS_HIP_CHECK(hipGetLastError());
    I->h_segbits.assign((size_t)I->Sw, 0);
    // everything that comes back lands in the context's pinned block first (results | decayed prior | segment bits | error word)
    const size_t rb_res = ((size_t)nB * sizeof(BaseOut) + 255) & ~(size_t)255, rb_cls = ((size_t)c->nS * 4 + 255) & ~(size_t)255, rb_seg = ((size_t)I->Sw * 4 + 255) & ~(size_t)255;
    { const int rcp = ensure_pinned(c, (size_t)PIN_VAR + rb_res + rb_cls + rb_seg + 256); if (rcp) return rcp; }
    char* rbp = (char*)c->h_pin + PIN_VAR;
    unsigned int* q_err_pin = (unsigned int*)(rbp + rb_res + rb_cls + rb_seg);
    *q_err_pin = 0u;
    STOCS_HIP_CHECK(hipMemcpyAsync(rbp, sb.res, (size_t)nB * sizeof(BaseOut), hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(rbp + rb_res, I->d_cls, (size_t)c->nS * 4, hipMemcpyDeviceToHost, c->stream));   // the decayed prior (Q8)
    STOCS_HIP_CHECK(hipMemcpyAsync(rbp + rb_res + rb_cls, I->d_segbits, (size_t)I->Sw * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(q_err_pin, A.q_err, 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<BaseOut> res((const BaseOut*)rbp, (const BaseOut*)rbp + nB);
    memcpy(c->h_sprob.data(), rbp + rb_res, (size_t)c->nS * 4);
    memcpy(I->h_segbits.data(), rbp + rb_res + rb_cls, (size_t)I->Sw * 4);
    const unsigned int q_err = *q_err_pin;
    if (q_err) { set_error("instance-mode sampling: the second workgroup gave up waiting for the first"); return STOCS_ERR_HIP; }
    if (dbg) {
        unsigned long long st[16];
        STOCS_HIP_CHECK(hipMemcpy(st, I->d_parent, 128, hipMemcpyDeviceToHost));
        const double per = 1.0 / std::max(nB, 1);
        fprintf(stderr, "[stocs instance] %d attempts, shader cycles per attempt (s_memtime).  first workgroup: weights + publish %.0f | draw 1 %.0f | pass 1 + max distance %.0f | flood fill %.0f | bookkeeping + compaction + hand-over %.0f.  second workgroup: waiting %.0f | points 2-4 %.0f | ordered bases (once, /attempt) %.0f\n",
                nB, st[0] * per, st[1] * per, st[2] * per, st[3] * per, st[4] * per, st[5] * per, st[6] * per, st[7] * per);
        fprintf(stderr, "[stocs instance] per attempt: %.0f of %d points within key distance of point 1 (stage A of pass 1: %.0f cycles), %.0f survivors inside the mask\n",
                st[8] * per, c->nS, st[10] * per, st[9] * per);
    }
    c->last_segment.clear();
    if (nB > 0 && res[(size_t)nB - 1].ids[0] >= 0)   // `segment` of the last attempt that got as far as its mask
        for (int i = 0; i < c->nS; ++i) if ((I->h_segbits[(size_t)(i >> 5)] >> (i & 31)) & 1u) c->last_segment.push_back(i);
    return record_bases(c, nB, res.data(), ids, inv, valid);
}

// ---------------------------------------------------------------------------------------------------------------
// Base sampling of a whole batch of independent trials in ONE launch (stocs_run_trials): trial t is what
// stocs_sample_bases(mode, seeds[t], 0, n_attempts) gives on a freshly reset context.
//   class mode: n_trials x n_attempts workgroups of class_attempts_kernel (the attempts are independent anyway);
//   instance mode: a pair of workgroups per trial (the attempts of ONE trial are sequential, stocs.cpp:572-580,626), every trial
//   on its own copy of the mutable state -- decaying prior, previous_segment, segmentation buffer, masks, hand-over slots -- and
//   with its own copy of the scene normals + LCP weights (Q8: the LCP adds the DECAYED class probability of its trial).
// res_host: n_trials * n_attempts results.  *snrmw0 / *snrmw_stride: trial t's weights for the scoring kernel (instance mode; NULL
// in class mode: the scene's own).  The context's own per-trial state is not touched.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_trial_state_kernel(char* __restrict__ block, size_t stride, size_t zero_words4, size_t o_cls, size_t o_snrm,
                                                               const float4* __restrict__ snrmw, int S) {
    char* mine = block + (size_t)blockIdx.y * stride;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < zero_words4) ((uint4*)mine)[i] = make_uint4(0u, 0u, 0u, 0u);   // prev_in | label | masks | segment | flags | error word
    if (i < (size_t)S) { const float4 v = snrmw[i]; ((float*)(mine + o_cls))[i] = v.w; ((float4*)(mine + o_snrm))[i] = v; }
}

int sample_trials(stocs_ctx* c, int mode, int nT, const uint64_t* seeds, int nA, float dispersion, BaseOut* res_host, const float4** snrmw0, size_t* snrmw_stride) {
    *snrmw0 = NULL; *snrmw_stride = 0;
    const size_t S = (size_t)c->nS, nW = (size_t)nT * (size_t)nA;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    if (mode == 0) {
        const bool wlds = S <= 26000 && !getenv("STOCS_INSTANCE_NO_LDS");
        // scenes beyond the LDS working set keep 8 bytes per (attempt, point) in device memory: at most ~1 GB of it per launch
        const size_t per_launch = wlds ? nW : std::max<size_t>(1, std::min<size_t>(nW, ((size_t)1 << 30) / (S * 8)));
        const size_t b_res = al(nW * sizeof(BaseOut)), b_seed = al((size_t)nT * 8), b_w = wlds ? 0 : al(per_launch * S * 4), b_slots = al(nW * 4);
        int rc = ensure_scratch(c, b_res + b_seed + 2 * b_w + b_slots);
        if (rc) return rc;
        if ((rc = ensure_pinned(c, (size_t)PIN_VAR + b_seed + b_res))) return rc;     // seeds up, every attempt's result down
        char* p = (char*)c->d_scratch;
        ClassArgs A;
        A.pa = pass_args(c);
        A.res = (BaseOut*)p;
        uint64_t* d_seeds = (uint64_t*)(p + b_res);
        A.w_g = (float*)(p + b_res + b_seed); A.sv_g = (int32_t*)(p + b_res + b_seed + b_w);
        A.draw_per_thread = 2; A.stamps = NULL;
        A.seeds = d_seeds; A.per_trial = nA; A.slot_list = NULL;
        memcpy((char*)c->h_pin + PIN_VAR, seeds, (size_t)nT * 8);
        STOCS_HIP_CHECK(hipMemcpyAsync(d_seeds, (char*)c->h_pin + PIN_VAR, (size_t)nT * 8, hipMemcpyHostToDevice, c->stream));
        const size_t lds = wlds ? ((S * 4 + 15) & ~(size_t)15) + S * 2 + 16 : 0;
        if (wlds) STOCS_HIP_CHECK(hipFuncSetAttribute((const void*)class_attempts_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        if (wlds && lds <= CLASS_TWO_LDS) STOCS_HIP_CHECK(hipFuncSetAttribute((const void*)class_attempts_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLASS_TWO_LDS));
        const bool lean = wlds && lean_usable(c);
        if (lean && (rc = ensure_prior_cdf(c))) return rc;
        for (size_t w0 = 0; w0 < nW; w0 += per_launch) {
            const unsigned n = (unsigned)std::min(per_launch, nW - w0);
            A.wg_offset = (int)w0;
            if (lean && lean_quarter(S)) hipLaunchKernelGGL(class_attempts_lean_kernel<256>, dim3(n), dim3(256), lean_lds_bytes(S), c->stream, A, (uint64_t)0, 0, (int)n, (const unsigned long long*)c->d_cdf, lean_cap(S));
            else if (lean && lean_half(S)) hipLaunchKernelGGL(class_attempts_lean_kernel<512>, dim3(n), dim3(512), lean_lds_bytes(S), c->stream, A, (uint64_t)0, 0, (int)n, (const unsigned long long*)c->d_cdf, lean_cap(S));
            else if (lean) hipLaunchKernelGGL(class_attempts_lean_kernel<1024>, dim3(n), dim3(1024), lean_lds_bytes(S), c->stream, A, (uint64_t)0, 0, (int)n, (const unsigned long long*)c->d_cdf, lean_cap(S));
            else if (wlds && lds <= CLASS_TWO_LDS && n > 256) hipLaunchKernelGGL((class_attempts_kernel<true, true>), dim3(n), dim3(1024), lds, c->stream, A, (uint64_t)0, 0, (int)n);
            else if (wlds) hipLaunchKernelGGL(class_attempts_kernel<true>, dim3(n), dim3(1024), lds, c->stream, A, (uint64_t)0, 0, (int)n);
            else hipLaunchKernelGGL(class_attempts_kernel<false>, dim3(n), dim3(1024), 0, c->stream, A, (uint64_t)0, 0, (int)n);
        }
        STOCS_HIP_CHECK(hipGetLastError());
        // (through the pinned block: 200 KB of results for 64 trials into the caller's pageable vector took the runtime's staging path)
        char* res_pin = (char*)c->h_pin + PIN_VAR + b_seed;
        STOCS_HIP_CHECK(hipMemcpyAsync(res_pin, A.res, nW * sizeof(BaseOut), hipMemcpyDeviceToHost, c->stream));
        STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
        memcpy(res_host, res_pin, nW * sizeof(BaseOut));
        if (lean) { A.wg_offset = 0; if ((rc = redo_lean_overflows(c, A, 0, 0, res_host, nW, (int32_t*)((char*)c->d_scratch + b_res + b_seed + 2 * b_w)))) return rc; }
        return STOCS_OK;
    }
    // ---- instance mode ----
    int rc = prepare_instance_state(c);
    if (rc) return rc;
    InstanceState* I = (InstanceState*)c->inst;
    const size_t Sw = (size_t)I->Sw, nr = std::max<size_t>(I->n_runs, 1), nq = (size_t)std::max(nA, 1);
    // one trial's block: what has to start from zero comes first (one fill), then everything that is written before it is read
    const size_t o_pi = 0, o_lb = o_pi + al(S), o_mb = o_lb + al(S), o_sb = o_mb + al(256 * Sw * 4), o_fl = o_sb + al(Sw * 4), o_er = o_fl + al(nq * 4), zero_end = o_er + 256,
                 o_cl = zero_end, o_pa = o_cl + al(S * 4), o_w = o_pa + al((nr + 1) * 4), o_sv = o_w + al(S * 4), o_sn = o_sv + al(S * 4), o_rs = o_sn + al(S * 16),
                 o_hd = o_rs + al(nq * sizeof(BaseOut)), o_qs = o_hd + al(nq * 16), o_qw = o_qs + al(nq * S * 4), stride = o_qw + al(nq * S * 4);
    const size_t b_seed = al((size_t)nT * 8), total = b_seed + (size_t)nT * stride;
    if (I->trials_bytes < total) {
        if (I->d_trials) { STOCS_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipFree(I->d_trials); I->d_trials = NULL; I->trials_bytes = 0; }
        STOCS_HIP_CHECK(dev_malloc((void**)&I->d_trials, total + total / 8));
        I->trials_bytes = total + total / 8;
    }
    if ((rc = ensure_pinned(c, (size_t)PIN_VAR + b_seed))) return rc;
    uint64_t* d_seeds = (uint64_t*)I->d_trials;
    char* blk = I->d_trials + b_seed;
    memcpy((char*)c->h_pin + PIN_VAR, seeds, (size_t)nT * 8);
    STOCS_HIP_CHECK(hipMemcpyAsync(d_seeds, (char*)c->h_pin + PIN_VAR, (size_t)nT * 8, hipMemcpyHostToDevice, c->stream));
    {
        const size_t zw = zero_end / 16, n = std::max(zw, S);
        hipLaunchKernelGGL(init_trial_state_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nT), dim3(256), 0, c->stream, blk, stride, zw, o_cl, o_sn,
                           (const float4*)c->d_snrmw, (int)S);
    }
    InstanceArgs A;
    A.pa = pass_args(c);
    A.pix = c->d_spix; A.edge_pt = I->d_edge_pt; A.pt_run = I->d_pt_run; A.run_s = I->d_run_s; A.run_e = I->d_run_e; A.row_off = I->d_row_off;
    A.pairs = I->d_pairs; A.pair_off = I->d_pair_off;
    A.draw_per_thread = 2;
    if (const char* e = getenv("STOCS_DRAW_PER_THREAD")) A.draw_per_thread = std::max(1, atoi(e));
    A.H = c->prm.image_height; A.W = c->prm.image_width; A.Sw = I->Sw;
    A.cls = (float*)(blk + o_cl); A.prev_in = (uint8_t*)(blk + o_pi); A.label = (uint8_t*)(blk + o_lb); A.maskbits = (uint32_t*)(blk + o_mb);
    A.segbits = (uint32_t*)(blk + o_sb); A.parent_g = (uint32_t*)(blk + o_pa);
    A.stamps = NULL;
    A.w = (float*)(blk + o_w); A.sv = (int32_t*)(blk + o_sv); A.spos_w = NULL; A.snrm_w = (float4*)(blk + o_sn); A.res = (BaseOut*)(blk + o_rs);
    A.q_hdr = (int4*)(blk + o_hd); A.q_flag = (unsigned int*)(blk + o_fl); A.q_err = (unsigned int*)(blk + o_er);
    A.q_sv = (int32_t*)(blk + o_qs); A.q_w = (float*)(blk + o_qw);
    A.n_trials = nT; A.trial_stride = stride; A.seeds = d_seeds;
    const size_t lds_parent = ((size_t)(INST_MAX_NODES + 1) * 4 + 15) & ~(size_t)15;
    const bool wlds = c->nS <= INST_LDS_POINTS && !getenv("STOCS_INSTANCE_NO_LDS");
    const size_t lds = lds_parent + (wlds ? (((S * 4 + 15) & ~(size_t)15) + S * 2 + 16) : 0);
    {
        const void* fn = wlds ? (const void*)instance_attempts_kernel<true> : (const void*)instance_attempts_kernel<false>;
        STOCS_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    }
    // The second workgroup of a trial polls the first's flags.  Correctness does not depend on the launch size: a consumer only waits
    // for the producer with the next-lower workgroup index, which the hardware has dispatched before it (workgroups start in index
    // order).  The launch is cut to one workgroup per CU of THIS device (a 1024-thread workgroup with this much LDS takes a whole
    // CU) so that every pair of a launch starts together instead of consumers idling on CUs that later producers are waiting for.
    int n_cu = 256;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0) n_cu = v; }
    const int per_launch = std::max(1, n_cu / 2);
    for (int t0 = 0; t0 < nT; t0 += per_launch) {
        const int n = std::min(per_launch, nT - t0);
        InstanceArgs B = A;
        B.n_trials = n; B.seeds = d_seeds + t0;
#define TR_ADV(p) B.p = (decltype(B.p))((char*)(A.p) + (size_t)t0 * stride)
        TR_ADV(cls); TR_ADV(prev_in); TR_ADV(label); TR_ADV(maskbits); TR_ADV(segbits); TR_ADV(parent_g); TR_ADV(w); TR_ADV(sv); TR_ADV(snrm_w); TR_ADV(res);
        TR_ADV(q_hdr); TR_ADV(q_sv); TR_ADV(q_w); TR_ADV(q_flag); TR_ADV(q_err);
#undef TR_ADV
        if (wlds) hipLaunchKernelGGL(instance_attempts_kernel<true>, dim3(2u * (unsigned)n), dim3(1024), lds, c->stream, B, (uint64_t)0, 0, nA, dispersion);
        else hipLaunchKernelGGL(instance_attempts_kernel<false>, dim3(2u * (unsigned)n), dim3(1024), lds, c->stream, B, (uint64_t)0, 0, nA, dispersion);
    }
    STOCS_HIP_CHECK(hipGetLastError());
    // results and error words of all trials: strided rows of the block
    std::vector<unsigned int> q_err((size_t)nT, 0u);
    STOCS_HIP_CHECK(hipMemcpy2DAsync(res_host, (size_t)nA * sizeof(BaseOut), blk + o_rs, stride, (size_t)nA * sizeof(BaseOut), (size_t)nT, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpy2DAsync(q_err.data(), 4, blk + o_er, stride, 4, (size_t)nT, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int t = 0; t < nT; ++t)
        if (q_err[(size_t)t]) { set_error("instance-mode sampling (trial %d of the batch): the second workgroup gave up waiting for the first", t); return STOCS_ERR_HIP; }
    *snrmw0 = (const float4*)(blk + o_sn);
    *snrmw_stride = stride;
    return STOCS_OK;
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_sample_bases(stocs_ctx* c, int mode, uint64_t seed, int first_attempt, int n_attempts, float dispersion,
                       int32_t* base_ids4, float* inv2, int32_t* valid) {
    if (!c || n_attempts < 0 || first_attempt < 0 || (mode != 0 && mode != 1)) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->index.built) { set_error("stocs_sample_bases: PPF index not built"); return STOCS_ERR_STATE; }
    clear_trial_batch(c);
    if (n_attempts == 0) return STOCS_OK;
    if (c->nS == 0) {  // the reference would index an empty vector here (stocs.cpp:386); report no bases
        for (int b = 0; b < n_attempts; ++b) if (valid) valid[b] = 0;
        return STOCS_OK;
    }
    if (mode == 0) return sample_class(c, seed, first_attempt, n_attempts, base_ids4, inv2, valid);
    if (n_attempts + first_attempt > 254) { set_error("instance mode labels segments with a u8 (<= 254 attempts, Q14)"); return STOCS_ERR_INVALID; }
    return sample_instance(c, seed, first_attempt, n_attempts, dispersion, base_ids4, inv2, valid);
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

void stocs_internal_free_instance(stocs_ctx* c) { if (c) free_instance_state(c); }
// the edge map or the scene changed: runs, per-point tables and the per-trial state are rebuilt at the next instance-mode call
void stocs_internal_invalidate_instance(stocs_ctx* c) { if (c && c->inst) ((InstanceState*)c->inst)->runs_valid = false; }

int stocs_reset_trial(stocs_ctx* c) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->h_sprob = c->h_sprob0;
    c->last_segment.clear();
    clear_trial_batch(c);
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1;
    int rc = reset_instance_trial(c);
    if (rc) return rc;
    return refresh_class_prob_on_device(c);
}

int stocs_set_bases(stocs_ctx* c, int n, const int32_t* ids, const float* inv) {
    if (!c || n < 0 || (n && (!ids || !inv))) return STOCS_ERR_INVALID;
    for (int i = 0; i < 4 * n; ++i)
        if (ids[i] < 0 || ids[i] >= c->nS) { set_error("stocs_set_bases: scene index out of range"); return STOCS_ERR_INVALID; }
    clear_trial_batch(c);
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
    clear_trial_batch(c);
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

// Device self-check of the float filter in front of the PPF arithmetic (ppf_key_fast): n_pairs seeded pairs of the
// context's scene points are keyed with the filter and with the reference's double arithmetic alone; *n_mismatch must be 0.
int stocs_ppf_filter_check(stocs_ctx* c, uint64_t seed, int64_t n_pairs, int64_t* n_tested, int64_t* n_undecided, int64_t* n_mismatch) {
    if (!c || n_pairs <= 0 || n_pairs > 0x7FFFFFFF || c->nS < 2) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    int rc = ensure_scratch(c, 256);
    if (rc) return rc;
    unsigned int* d_counts = (unsigned int*)c->d_scratch;
    STOCS_HIP_CHECK(hipMemsetAsync(d_counts, 0, 16, c->stream));
    PassArgs a = pass_args(c);
    // no index needed for this check: the key space of the parameters, as wide in distance as the packed key allows
    a.ix.tr = c->prm.ppf_tr_discretization; a.ix.rot = c->prm.ppf_rot_discretization;
    a.ix.NA = 180 / a.ix.rot + 1;
    a.ix.nD = (int)std::min<uint64_t>(16384, (0xFFFFFFFFull / ((uint64_t)a.ix.NA * a.ix.NA * a.ix.NA)) - 1);
    set_distance_thresholds(&a.ix);
    hipLaunchKernelGGL(ppf_filter_check_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, c->stream, a, seed, (uint32_t)n_pairs, d_counts);
    STOCS_HIP_CHECK(hipGetLastError());
    unsigned int h[4] = {0, 0, 0, 0};
    STOCS_HIP_CHECK(hipMemcpyAsync(h, d_counts, 16, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (n_tested) *n_tested = h[0];
    if (n_undecided) *n_undecided = h[1];
    if (n_mismatch) *n_mismatch = h[2];
    return STOCS_OK;
}

int stocs_weight_fix_check(stocs_ctx* c, int64_t* n_mismatch) {
    if (!c || !n_mismatch) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    int rc = ensure_scratch(c, 256);
    if (rc) return rc;
    unsigned int* d_counts = (unsigned int*)c->d_scratch;
    STOCS_HIP_CHECK(hipMemsetAsync(d_counts, 0, 16, c->stream));
    hipLaunchKernelGGL(weight_fix_check_kernel, dim3(1u << 16), dim3(256), 0, c->stream, d_counts);   // 2^16 * 2^8 threads * 2^8 patterns
    STOCS_HIP_CHECK(hipGetLastError());
    unsigned int h = 0;
    STOCS_HIP_CHECK(hipMemcpyAsync(&h, d_counts, 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    *n_mismatch = h;
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
