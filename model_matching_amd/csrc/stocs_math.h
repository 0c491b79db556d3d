// stocs_math.h -- float vector algebra and deterministic transcendental helpers shared by the host
// (C++) and device (HIP, gfx950) sides of the product.  NOT shared with oracle/ (the oracle has its
// own restatement on top of libm).
//
// Numerics contract (DESIGN.md "numerics"):
//   * every translation unit is built with -ffp-contract=off; divide and sqrt are IEEE;
//   * 3-term sums are e0 + (e1 + e2)   (Eigen redux order used by dot/squaredNorm/3x3 products);
//   * 4x4 * homogeneous(vec3) is ((m0*x + m1*y) + m2*z) + m3;
//   * normalized(v): z = |v|^2; z > 0 ? v / sqrt(z) : v.
// With these rules the host, the device and the oracle produce bit-identical floats for everything
// that does not go through libm.  The only per-point transcendental on the hot path is the PPF
// angle  atan2(|n x u|, n.u)  (reference src/rgbd.cpp:113-115): it is evaluated in double by
// stocs_atan2() below -- plain + - * / on doubles, so host and device agree exactly, and it agrees
// with glibc's double atan2 (<= 1 ulp each) unless the true angle is within ~1e-14 of an integer
// degree.  The acos-based predicates (stocs.cpp:428-440, 1028-1032) are turned into exact float
// thresholds on the dot product (compute_thresholds in ctx.hip).
#ifndef STOCS_MATH_H
#define STOCS_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define STOCS_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define STOCS_HD inline
#endif

namespace stocs {

struct V3 {
    float x, y, z;
};
STOCS_HD V3 mk3(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
STOCS_HD V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
STOCS_HD V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
STOCS_HD V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
STOCS_HD V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
STOCS_HD V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
STOCS_HD V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
STOCS_HD float dot3(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
STOCS_HD float sqn3(V3 a) { return dot3(a, a); }
// sqrtf lowers to v_sqrt_f32 + two FMA correction steps (correctly rounded) on gfx950 under the
// default -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn lowers to the bare 1-ulp v_sqrt_f32
// and must NOT be used here.
STOCS_HD float stocs_sqrtf(float x) { return sqrtf(x); }
STOCS_HD float norm3(V3 a) { return stocs_sqrtf(sqn3(a)); }
STOCS_HD V3 cross3(V3 a, V3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
STOCS_HD V3 normalized3(V3 a) {
    float z = sqn3(a);
    if (z > 0.0f) return a / stocs_sqrtf(z);
    return a;
}
STOCS_HD float comp3(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// column-major 4x4 (Eigen::Matrix4f::data() layout): T[c*4+r]
STOCS_HD V3 xform_point(const float* T, V3 p) {
    return mk3(((T[0] * p.x + T[4] * p.y) + T[8] * p.z) + T[12],
               ((T[1] * p.x + T[5] * p.y) + T[9] * p.z) + T[13],
               ((T[2] * p.x + T[6] * p.y) + T[10] * p.z) + T[14]);
}
STOCS_HD V3 xform_normal(const float* T, V3 n) {
    return mk3(T[0] * n.x + (T[4] * n.y + T[8] * n.z),
               T[1] * n.x + (T[5] * n.y + T[9] * n.z),
               T[2] * n.x + (T[6] * n.y + T[10] * n.z));
}

// ---------------------------------------------------------------------------------------------
// deterministic double atan2 for y >= 0 or any sign; basic IEEE operations only.
// atan(r), r in [0,1]:  c = k/8 (k = round(8r)), t = (r - c)/(1 + r c), atan r = atan c + atan t,
// |t| <= 1/16, odd Taylor series through t^13 (next term < 6e-20).
// ---------------------------------------------------------------------------------------------
STOCS_HD double stocs_atan01(double r) {
    const double TAB[9] = {0.0,
                           0.12435499454676144,
                           0.24497866312686414,
                           0.35877067027057225,
                           0.4636476090008061,
                           0.5585993153435624,
                           0.6435011087932844,
                           0.7188299996216245,
                           0.7853981633974483};
    int k = (int)(r * 8.0 + 0.5);
    if (k > 8) k = 8;
    double c = (double)k * 0.125;
    double t = (r - c) / (1.0 + r * c);
    double t2 = t * t;
    double s = 1.0 / 13.0;
    s = -1.0 / 11.0 + t2 * s;
    s = 1.0 / 9.0 + t2 * s;
    s = -1.0 / 7.0 + t2 * s;
    s = 1.0 / 5.0 + t2 * s;
    s = -1.0 / 3.0 + t2 * s;
    s = 1.0 + t2 * s;
    // select instead of indexing so that no scratch array is needed on the device
    double base = TAB[0];
    base = (k == 1) ? TAB[1] : base;
    base = (k == 2) ? TAB[2] : base;
    base = (k == 3) ? TAB[3] : base;
    base = (k == 4) ? TAB[4] : base;
    base = (k == 5) ? TAB[5] : base;
    base = (k == 6) ? TAB[6] : base;
    base = (k == 7) ? TAB[7] : base;
    base = (k == 8) ? TAB[8] : base;
    return base + t * s;
}

STOCS_HD double stocs_atan2(double y, double x) {
    const double PI = 3.141592653589793, PI_2 = 1.5707963267948966;
    if (!(y == y) || !(x == x)) return y + x;  // NaN
    const bool yneg = (y < 0.0) || (y == 0.0 && 1.0 / y < 0.0);
    const bool xneg = (x < 0.0) || (x == 0.0 && 1.0 / x < 0.0);
    const double ay = yneg ? -y : y;
    const double ax = xneg ? -x : x;
    double a;
    if (ay == 0.0) {
        a = xneg ? PI : 0.0;
    } else if (ax == 0.0) {
        a = PI_2;
    } else {
        if (ay > ax) {
            a = PI_2 - stocs_atan01(ax / ay);
        } else {
            a = stocs_atan01(ay / ax);
        }
        if (xneg) a = PI - a;
    }
    return yneg ? -a : a;
}

// int(double) with the x86 cvttsd2si convention for NaN / out of range (INT_MIN): such PPF keys are
// never present in the index.
STOCS_HD int stocs_trunc_int(double d) {
    if (!(d > -2147483648.0 && d < 2147483648.0)) return (int)0x80000000;
    return (int)d;
}

// reference src/rgbd.cpp:85-97
STOCS_HD int ppf_closest_bin(int value, int discretization) {
    int lower_limit = value - (value % discretization);
    int upper_limit = lower_limit + discretization;
    int dist_from_lower = value - lower_limit;
    int dist_from_upper = upper_limit - value;
    return (dist_from_lower < dist_from_upper) ? lower_limit : upper_limit;
}

// reference src/rgbd.cpp:99-121 (u = p1 - p2 for both normals; angle in degrees = a*180/M_PI in double)
STOCS_HD void ppf_compute(V3 p1, V3 n1, V3 p2, V3 n2, int tr, int rot, int* out4) {
    const double PI = 3.14159265358979323846;
    V3 u = p1 - p2;
    int f0 = stocs_trunc_int((double)(norm3(u) * 1000.0f));
    int f1 = stocs_trunc_int(stocs_atan2((double)norm3(cross3(n1, u)), (double)dot3(n1, u)) * 180.0 / PI);
    int f2 = stocs_trunc_int(stocs_atan2((double)norm3(cross3(n2, u)), (double)dot3(n2, u)) * 180.0 / PI);
    int f3 = stocs_trunc_int(stocs_atan2((double)norm3(cross3(n1, n2)), (double)dot3(n1, n2)) * 180.0 / PI);
    out4[0] = ppf_closest_bin(f0, tr);
    out4[1] = ppf_closest_bin(f1, rot);
    out4[2] = ppf_closest_bin(f2, rot);
    out4[3] = ppf_closest_bin(f3, rot);
}

// PPF key space: distance bin index d = K0/tr in [0, nD), angle bin index a = Ki/rot in [0, NA)
// with NA = 180/rot + 1.  Packed key = ((d*NA + a1)*NA + a2)*NA + a3.
STOCS_HD uint32_t ppf_pack(int d, int a1, int a2, int a3, int NA) {
    return (uint32_t)(((d * NA + a1) * NA + a2) * NA + a3);
}

// seeded draw that replaces the clock-seeded std::discrete_distribution of stocs.cpp:133-148
// (documented divergence Q6): splitmix-style counter RNG + 2^32 fixed-point weights.
STOCS_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
STOCS_HD uint64_t rng64(uint64_t seed, uint64_t attempt, uint64_t k) {
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull);
    z = mix64(z ^ (attempt * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull));
    z = mix64(z ^ ((k + 1) * 0xDB4F0B9175AE2165ull));
    return z;
}
// trunc(w * 2^32) for w > 0 (0 for w <= 0 and NaN, saturating at 2^64 - 1) -- the value of
//     (uint64_t)((double)w * 4294967296.0)
// read off the bits of w: the product is exact, so the result is the 24-bit significand shifted by the exponent.  Equal
// to the double formula for all 2^32 floats (tests/test_abi_cpu.py checks a dense sample); no double arithmetic and no
// 64-bit conversion on the device.
STOCS_HD uint64_t weight_fix(float w) {
    union { float f; uint32_t u; } cv;
    cv.f = w;
    const uint32_t b = cv.u;
    if ((int32_t)b <= 0) return 0;                         // sign bit set, or +0
    const int e = (int)(b >> 23);                          // biased exponent
    if (e == 255) return (b & 0x7FFFFFu) ? 0 : 0xFFFFFFFFFFFFFFFFull;   // NaN / +inf
    if (e == 0) return 0;                                  // denormal: below 2^-126
    const uint64_t m = (uint64_t)((b & 0x7FFFFFu) | 0x800000u);
    const int sh = e - 118;                                // (e - 127) - 23 + 32
    if (sh >= 41) return 0xFFFFFFFFFFFFFFFFull;
    if (sh >= 0) return m << sh;
    if (sh <= -24) return 0;
    return m >> (-sh);
}
STOCS_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

}  // namespace stocs
#endif
