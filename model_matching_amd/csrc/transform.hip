// transform.hip -- candidate rigid transforms from (base, congruent quad) pairs, and the
// verification driver.  Replaces
//   ComputeRigidTransformation                         reference src/stocs.cpp:270-361
//   stocs_estimator::get_rigid_transform_from_congruent_pair      stocs.cpp:871-941
//   the <=200-per-base loop of run_stocs_estimation     src/stocs_match_one_object.cpp:120-147
//   stocs_estimator::compute_best_transform             stocs.cpp:982-1004
// One thread per (base, quad): three-point frame alignment (no SVD, no MFMA: a 3x3*3x3 product per
// candidate is not a dense contraction).  Pure IEEE float arithmetic in the order fixed by
// stocs_math.h, so the result is bit-identical to the CPU restatement.
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <cstring>


#include <algorithm>
#include <unordered_map>

#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

struct M3 { float m[3][3]; };
STOCS_HD M3 mul33(const M3& A, const M3& B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C.m[i][j] = A.m[i][0] * B.m[0][j] + (A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j]);
    return C;
}
STOCS_HD V3 mul3v(const M3& A, V3 v) {
    return mk3(A.m[0][0] * v.x + (A.m[0][1] * v.y + A.m[0][2] * v.z),
               A.m[1][0] * v.x + (A.m[1][1] * v.y + A.m[1][2] * v.z),
               A.m[2][0] * v.x + (A.m[2][1] * v.y + A.m[2][2] * v.z));
}
__device__ __forceinline__ V3 ld3(const float4* a, int i) { float4 v = a[i]; return mk3(v.x, v.y, v.z); }

struct XformJob { int32_t s[4]; int32_t q[4]; };  // scene base ids, model quad ids

// One candidate: the arithmetic of ComputeRigidTransformation + get_rigid_transform_from_congruent_pair on three point
// pairs.  Host and device run this same function (IEEE float operations in a fixed order, sqrt the only non-trivial one),
// so a candidate computed by the per-call entry point on the host equals the batched kernel's bit for bit.
STOCS_HD int rigid_transform_one(V3 p0, V3 p1, V3 p2, V3 q0, V3 q1, V3 q2, V3 cscene, V3 cmodel, float* T, float* P) {
    const V3 centroid1 = ((p0 + p1) + p2) / 3.0f;  // stocs.cpp:885
    const V3 centroid2 = ((q0 + q1) + q2) / 3.0f;  // stocs.cpp:907-909
    int ok = 1;
    // degenerate frames: the reference returns kLargeNumber from a bool function (true with an
    // uninitialised matrix, Q2) -- deliberate divergence: reject.
    V3 vp1 = p1 - p0;
    if (sqn3(vp1) == 0) ok = 0;
    vp1 = normalized3(vp1);
    V3 vp2 = (p2 - p0) - (dot3(p2 - p0, vp1) * vp1);
    if (sqn3(vp2) == 0) ok = 0;
    vp2 = normalized3(vp2);
    const V3 vp3 = cross3(vp1, vp2);
    V3 vq1 = q1 - q0;
    if (sqn3(vq1) == 0) ok = 0;
    vq1 = normalized3(vq1);
    V3 vq2 = (q2 - q0) - (dot3(q2 - q0, vq1) * vq1);
    if (sqn3(vq2) == 0) ok = 0;
    vq2 = normalized3(vq2);
    const V3 vq3 = cross3(vq1, vq2);
    // rotation = rotate_p.transpose() * rotate_q with the frames as rows (stocs.cpp:316-326)
    M3 Pt, Q;
    Pt.m[0][0] = vp1.x; Pt.m[1][0] = vp1.y; Pt.m[2][0] = vp1.z;
    Pt.m[0][1] = vp2.x; Pt.m[1][1] = vp2.y; Pt.m[2][1] = vp2.z;
    Pt.m[0][2] = vp3.x; Pt.m[1][2] = vp3.y; Pt.m[2][2] = vp3.z;
    Q.m[0][0] = vq1.x; Q.m[0][1] = vq1.y; Q.m[0][2] = vq1.z;
    Q.m[1][0] = vq2.x; Q.m[1][1] = vq2.y; Q.m[1][2] = vq2.z;
    Q.m[2][0] = vq3.x; Q.m[2][1] = vq3.y; Q.m[2][2] = vq3.z;
    const M3 R = mul33(Pt, Q);
    // (rotation*rotation).diagonal() - 1 > 1e-6 (sic: R*R not R*R^T, Q3; stocs.cpp:329)
    const M3 RR = mul33(R, R);
    const float kSmall = 1e-6f;
    if ((RR.m[0][0] - 1.0f > kSmall) || (RR.m[1][1] - 1.0f > kSmall) || (RR.m[2][2] - 1.0f > kSmall)) ok = 0;
    // rms (stocs.cpp:334-346) only gates through "rms >= 0" (:922): false iff NaN
    {
        float rms = 0.0f;
        rms += norm3((mul3v(R, 1.0f * q0 - centroid2) - p0) + centroid1);
        rms += norm3((mul3v(R, 1.0f * q1 - centroid2) - p1) + centroid1);
        rms += norm3((mul3v(R, 1.0f * q2 - centroid2) - p2) + centroid1);
        rms /= 4.0f;
        if (!(rms >= 0.0f)) ok = 0;
    }
    // etrans = I; scale(1); translate(c1); rotate(R); translate(-c2)  (stocs.cpp:348-357)
    const V3 t = centroid1 + mul3v(R, -centroid2);
    // camera-frame translation (stocs.cpp:925-933); rot*scale of computeRotationScaling == linear part
    const V3 tc = (centroid1 + cscene) - mul3v(R, centroid2 + cmodel);
#pragma unroll
    for (int col = 0; col < 3; ++col) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { T[col * 4 + r] = R.m[r][col]; P[col * 4 + r] = R.m[r][col]; }
        T[col * 4 + 3] = 0.0f; P[col * 4 + 3] = 0.0f;
    }
    T[12] = t.x; T[13] = t.y; T[14] = t.z; T[15] = 1.0f;
    P[12] = tc.x; P[13] = tc.y; P[14] = tc.z; P[15] = 1.0f;
    return ok;
}

__global__ __launch_bounds__(256) void rigid_transform_kernel(const float4* __restrict__ spos, const float4* __restrict__ mpos,
                                                              const XformJob* __restrict__ jobs, int n, V3 cscene, V3 cmodel,
                                                              float* __restrict__ T_out, float* __restrict__ P_out,
                                                              int32_t* __restrict__ ok_out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) ok_out[n] = 0;   // the flag array is scanned over n + 1 entries so that the last output is the accepted count
    if (j >= n) return;
    const XformJob job = jobs[j];
    const V3 p0 = ld3(spos, job.s[0]), p1 = ld3(spos, job.s[1]), p2 = ld3(spos, job.s[2]);
    const V3 q0 = ld3(mpos, job.q[0]), q1 = ld3(mpos, job.q[1]), q2 = ld3(mpos, job.q[2]);
    float T[16], P[16];
    const int ok = rigid_transform_one(p0, p1, p2, q0, q1, q2, cscene, cmodel, T, P);
    float4* To = (float4*)(T_out + (size_t)j * 16);
    float4* Po = (float4*)(P_out + (size_t)j * 16);
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        To[col] = make_float4(T[col * 4], T[col * 4 + 1], T[col * 4 + 2], T[col * 4 + 3]);
        Po[col] = make_float4(P[col * 4], P[col * 4 + 1], P[col * 4 + 2], P[col * 4 + 3]);
    }
    ok_out[j] = ok;
}

// the winner's camera-frame pose next to its key: one copy and one synchronisation tell the host everything
__global__ __launch_bounds__(64) void winner_pose_kernel(const unsigned long long* __restrict__ key, const float* __restrict__ P, int n, float* __restrict__ out18) {
    const unsigned long long k = *key;
    if (threadIdx.x == 0) { out18[0] = __uint_as_float((uint32_t)(k & 0xFFFFFFFFull)); out18[1] = __uint_as_float((uint32_t)(k >> 32)); }
    if (threadIdx.x < 16) {
        const uint32_t id = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull);
        out18[2 + threadIdx.x] = (k && id < (uint32_t)n) ? P[(size_t)id * 16 + threadIdx.x] : 0.0f;
    }
}

// The picks of stocs_make_transforms on the device, one workgroup per base: a base with fewer quads than the per-base maximum is
// used whole (ranks 0 .. nq-1 of its sorted run); a larger one gets the seeded sample without replacement -- the same sparse
// partial Fisher-Yates the host form runs (open-addressing table of the touched entries of the identity permutation, here in
// LDS): the draws r_j are independent and computed by all threads, the swaps are sequential and done by one.  Runs on the
// auxiliary stream next to the materialisation of the small bases.  table[b] = (first job, quad count lo, hi, unused).
// trial (when != NULL; stocs_run_trials): per base (seed lo, seed hi, slot of the base in ITS trial, 0) -- the draw of a base is
// rng(seed of its trial, its slot there, j), and its candidates carry that slot, exactly as when the trial runs alone.
__global__ __launch_bounds__(256) void draw_picks_kernel(const uint4* __restrict__ table, const uint4* __restrict__ trial, uint64_t seed, int max_per_base, uint32_t hmask,
                                                         int4* __restrict__ picks, int32_t* __restrict__ job_base) {
    extern __shared__ uint32_t lds_dyn[];
    int* hkeys = (int*)lds_dyn;                          // hmask + 1
    int* hvals = hkeys + (hmask + 1);                    // hmask + 1
    int* out = hvals + (hmask + 1);                      // max_per_base
    uint64_t* r = (uint64_t*)(out + ((max_per_base + 1) & ~1));   // max_per_base (8-byte aligned: everything before is an even number of words)
    const int b = blockIdx.x;
    const uint4 t = table[b];
    const int first = (int)t.x;
    const long long nq = (long long)(((unsigned long long)t.z << 32) | (unsigned long long)t.y);
    if (nq <= 0) return;
    int b_own = b;                                   // the base's slot in its own trial
    if (trial) { const uint4 tr = trial[b]; seed = ((uint64_t)tr.y << 32) | (uint64_t)tr.x; b_own = (int)tr.z; }
    if (nq < max_per_base) {   // stocs_match_one_object.cpp:126: strictly fewer -> all, in the std::set order of stocs.cpp:860-866
        for (int i = threadIdx.x; i < (int)nq; i += blockDim.x) { picks[first + i] = make_int4(b, i, first + i, 1); job_base[first + i] = b_own; }
        return;
    }
    for (uint32_t h = threadIdx.x; h <= hmask; h += blockDim.x) hkeys[h] = -1;
    for (int j = threadIdx.x; j < max_per_base; j += blockDim.x) r[j] = rng64(seed, 0x5E1EC7ull + (uint64_t)b_own, (uint64_t)j);
    __syncthreads();
    if (threadIdx.x == 0) {
        auto slot_of = [&](int i) {
            uint32_t h = ((uint32_t)i * 2654435761u) & hmask;
            while (hkeys[h] != -1 && hkeys[h] != i) h = (h + 1) & hmask;
            return h;
        };
        for (int j = 0; j < max_per_base; ++j) {
            const int k = j + (int)mulhi64(r[j], (uint64_t)(nq - j));
            const uint32_t hj = slot_of(j);
            const int vj = hkeys[hj] == j ? hvals[hj] : j;
            int vk = vj;
            if (k != j) {
                const uint32_t hk = slot_of(k);
                vk = hkeys[hk] == k ? hvals[hk] : k;
                hkeys[hk] = k; hvals[hk] = vj;
            }
            // (position j is never read again: steps go upward and k >= j -- the host form stores it all the same, with the same result)
            out[j] = vk;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < max_per_base; j += blockDim.x) { picks[first + j] = make_int4(b, out[j], first + j, 0); job_base[first + j] = b_own; }
}

// out[k] = a[idx[k]]: the accepted-candidate counts in front of every trial's first job (= where its candidates start)
__global__ __launch_bounds__(256) void gather_i32_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ idx, int n, int32_t* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = a[idx[k]];
}

// accepted candidates (ok != 0) keep their pick order: destination = exclusive scan of the flags
__global__ __launch_bounds__(256) void compact_candidates_kernel(const float4* __restrict__ T, const float4* __restrict__ P, const int32_t* __restrict__ ok,
                                                                 const int32_t* __restrict__ pos, const int32_t* __restrict__ job_base, int n,
                                                                 float4* __restrict__ To, float4* __restrict__ Po, float* __restrict__ lcp,
                                                                 int32_t* __restrict__ base_out, unsigned long long* __restrict__ best) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && best) *best = 0ull;   // the arg-max word of the verification that follows (stocs_verify_all): no fill of its own there
    if (j >= n || !ok[j]) return;
    const int d = pos[j];
#pragma unroll
    for (int k = 0; k < 4; ++k) { To[(size_t)d * 4 + k] = T[(size_t)j * 4 + k]; Po[(size_t)d * 4 + k] = P[(size_t)j * 4 + k]; }
    lcp[d] = 0.0f;   // "score is not computed at this time" (stocs.cpp:935-936)
    base_out[d] = job_base[j];
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_rigid_transform(stocs_ctx* c, const int32_t* ids4, const int32_t* quad4, float* T16, float* pose16, int* ok) {
    if (!c || !ids4 || !quad4 || !ok) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    for (int k = 0; k < 4; ++k)
        if (ids4[k] < 0 || ids4[k] >= c->nS || quad4[k] < 0 || quad4[k] >= c->nM) { set_error("index out of range"); return STOCS_ERR_INVALID; }
    // on the host, from the context's own copies of the centred clouds: the reference's caller asks for one candidate per
    // call, up to 200 per base (stocs_match_one_object.cpp:120-147) -- a device round trip each would cost 35 us
    float T[16], P[16];
    *ok = rigid_transform_one(c->h_spos[ids4[0]], c->h_spos[ids4[1]], c->h_spos[ids4[2]], c->h_mpos[quad4[0]], c->h_mpos[quad4[1]], c->h_mpos[quad4[2]],
                              c->centroid_scene, c->centroid_model, T, P);
    if (T16) memcpy(T16, T, 64);
    if (pose16) memcpy(pose16, P, 64);
    return STOCS_OK;
}

int stocs_make_transforms(stocs_ctx* c, int max_per_base, uint64_t seed, int* n_candidates) {
    if (!c || max_per_base <= 0) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (c->quad_off.size() != c->bases.size() + 1) { set_error("stocs_make_transforms: call stocs_find_congruent_all first"); return STOCS_ERR_STATE; }
    const bool dbg = getenv("STOCS_DEBUG_TIMING") != NULL;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    c->timing[1].begin();
    auto tick = [&](const char* label) {
        if (!dbg) return;
        (void)hipStreamSynchronize(c->stream);
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "[stocs transforms] %-18s %8.3f ms\n", label, (t1.tv_sec - ts0.tv_sec) * 1e3 + (t1.tv_nsec - ts0.tv_nsec) * 1e-6);
        ts0 = t1;
    };
    // A trial without a single congruent set -- no bases, empty pair lists, or no (base, cell) that both lists occupy -- is a valid,
    // empty result (the reference's loop simply appends nothing, stocs_match_one_object.cpp:111-147): no candidates, "no pose".
    if (c->quad_off.back() == 0) {
        clear_candidates(c);
        c->best_lcp = 0; c->best_index = -1;
        if (n_candidates) *n_candidates = 0;
        c->trial_cand_off.assign(c->trial_first_base.size(), 0);
        c->timing[1].lap("no congruent sets");
        return STOCS_OK;
    }
    // picks = (base, rank, job slot, sorted?) records; the quads themselves are produced on the device
    std::vector<int32_t> picks;
    std::vector<int> job_base;
    if (max_per_base > (1 << 24)) { set_error("stocs_make_transforms: max_per_base %d exceeds 2^24", max_per_base); return STOCS_ERR_INVALID; }
    uint32_t hsize = 64;
    while (hsize < 4u * (uint32_t)max_per_base) hsize <<= 1;
    const uint32_t hmask = hsize - 1;
    // the picks are drawn on the device (draw_picks_kernel, on the auxiliary stream next to the small bases' materialisation)
    // while the table of touched entries fits LDS; the host form below is the same draw and serves larger per-base maxima
    const bool device_picks = max_per_base <= 1024 && c->aux_stream && !getenv("STOCS_TRANSFORMS_HOST_PICKS");
    // host-drawn picks: the device starts on the small bases (used whole: materialised and sorted) while the host draws the subsets
    // of the large ones; device-drawn picks: behind the upload of the pick table, further down
    if (!device_picks && !c->bases.empty()) { const int rc0 = stocs_internal_prepare_small(c, max_per_base); if (rc0) return rc0; }
    // a batch of trials (stocs_run_trials): every base draws with the seed of its trial and under its slot there
    const bool batch = !c->base_seed.empty();
    if (batch && (c->base_seed.size() != c->bases.size() || c->base_local.size() != c->bases.size())) { set_error("internal: trial tables do not match the base set"); return STOCS_ERR_STATE; }
    const size_t nbases = c->bases.size(), tab_bytes = (16 * nbases + 255) & ~(size_t)255;
    const size_t n_tr = batch ? c->trial_first_base.size() : 0;     // trials + 1
    std::vector<size_t> first_job(nbases + 1, 0);
    size_t n_dev = 0;
    if (device_picks) {
        int rc0 = ensure_pinned(c, (size_t)PIN_VAR + 2 * tab_bytes + 4 * n_tr + 256);
        if (rc0) return rc0;
        uint4* table = (uint4*)((char*)c->h_pin + PIN_VAR);
        uint4* table2 = (uint4*)((char*)c->h_pin + PIN_VAR + tab_bytes);
        for (size_t b = 0; b < nbases; ++b) {
            const unsigned long long nq64 = c->quad_off[b + 1] - c->quad_off[b];
            if (nq64 > 0x7FFFFFFFull) { set_error("base %zu has %llu congruent quads (more than 2^31 - 1)", b, nq64); return STOCS_ERR_CAPACITY; }
            table[b] = make_uint4((uint32_t)n_dev, (uint32_t)nq64, (uint32_t)(nq64 >> 32), 0u);
            if (batch) table2[b] = make_uint4((uint32_t)c->base_seed[b], (uint32_t)(c->base_seed[b] >> 32), (uint32_t)c->base_local[b], 0u);
            first_job[b] = n_dev;
            n_dev += (size_t)std::min<unsigned long long>(nq64, (unsigned long long)max_per_base);
        }
        first_job[nbases] = n_dev;
    }
    std::vector<int> hkeys(hsize), hvals(hsize);
    auto add_pick = [&](size_t b, int rank, int sorted) {
        picks.push_back((int32_t)b); picks.push_back((int32_t)rank); picks.push_back((int32_t)job_base.size()); picks.push_back(sorted);
        job_base.push_back(batch ? (int)c->base_local[b] : (int)b);
    };
    for (size_t b = 0; b < c->bases.size() && !device_picks; ++b) {
        first_job[b] = job_base.size();
        const uint64_t seed_b = batch ? c->base_seed[b] : seed;
        const uint64_t slot_b = batch ? (uint64_t)c->base_local[b] : (uint64_t)b;
        const unsigned long long nq64 = c->quad_off[b + 1] - c->quad_off[b];
        if (nq64 > 0x7FFFFFFFull) { set_error("base %zu has %llu congruent quads (more than 2^31 - 1)", b, nq64); return STOCS_ERR_CAPACITY; }
        const long long nq = (long long)nq64;
        if (nq < max_per_base) {  // stocs_match_one_object.cpp:126: strictly fewer -> all, in the std::set order of stocs.cpp:860-866
            for (long long i = 0; i < nq; ++i) add_pick(b, (int)i, 1);
        } else {
            // seeded sample without replacement (divergence Q5 from the biased 2N-vector shuffle of
            // stocs_match_one_object.cpp:134-142, whose result depends on the C library's unseeded generator):
            // partial Fisher-Yates over the base's quads in EMISSION order (the order the loop of
            // stocs.cpp:827-858 finds them), kept sparse (only the touched entries of the identity
            // permutation are stored).  Any fixed enumeration serves a uniform draw; this one needs no sort.
            // sparse identity permutation: open-addressing table, reset per base (<= 2 * max_per_base live keys)
            std::fill(hkeys.begin(), hkeys.end(), -1);
            auto slot_of = [&](int i) {
                uint32_t h = ((uint32_t)i * 2654435761u) & hmask;
                while (hkeys[h] != -1 && hkeys[h] != i) h = (h + 1) & hmask;
                return h;
            };
            auto at = [&](int i) { const uint32_t h = slot_of(i); return hkeys[h] == i ? hvals[h] : i; };
            auto put = [&](int i, int v) { const uint32_t h = slot_of(i); hkeys[h] = i; hvals[h] = v; };
            for (int j = 0; j < max_per_base; ++j) {
                const uint64_t r = rng64(seed_b, 0x5E1EC7ull + slot_b, (uint64_t)j);
                const int k = j + (int)mulhi64(r, (uint64_t)(nq - j));
                const int vj = at(j), vk = at(k);
                put(j, vk); put(k, vj);
                add_pick(b, vk, 0);
            }
        }
    }
    if (!device_picks) first_job[nbases] = job_base.size();
    const size_t n = device_picks ? n_dev : job_base.size();
    c->timing[1].lap(device_picks ? "pick table (host)" : "small bases enqueued + host picks");
    tick("host picks");
    clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1;
    if (n) {
        // candidates stay on the device: jobs -> transforms -> order-preserving compaction of the accepted ones
        const size_t jb = ((n * sizeof(XformJob) + 255) / 256) * 256, tb = n * 64, ob = (((n + 1) * 4 + 255) / 256) * 256;
        size_t scan_tmp = 0;
        STOCS_HIP_CHECK(exclusive_scan(NULL, scan_tmp, (const uint32_t*)NULL, (uint32_t*)NULL, n + 1, c->stream));
        scan_tmp = ((scan_tmp + 255) / 256) * 256;
        const size_t pb = device_picks ? ((n * 16 + 255) / 256) * 256 + 2 * tab_bytes : 0;   // picks + the two tables
        const size_t trb = batch ? ((8 * n_tr + 255) / 256) * 256 : 0;                        // first jobs of the trials + their candidate offsets
        int rc = ensure_scratch(c, jb + 2 * tb + 3 * ob + scan_tmp + pb + trb);
        if (rc) return rc;
        if (!device_picks && (rc = ensure_pinned(c, (size_t)PIN_VAR + 2 * tab_bytes + 4 * n_tr + 256))) return rc;
        if ((size_t)c->cand_cap < n) {
            if (c->d_cand) { STOCS_HIP_CHECK(hipStreamSynchronize(c->stream)); (void)hipFree(c->d_cand); c->d_cand = NULL; }
            c->cand_cap = (int)(n + n / 4 + 1024);
            c->cand_bytes = (size_t)c->cand_cap * (16 + 16 + 1 + 1) * 4;
            STOCS_HIP_CHECK(dev_malloc((void**)&c->d_cand, c->cand_bytes));
        }
        c->timing[1].lap("buffers (scratch, pinned, candidate block)");
        if (!c->d_best) STOCS_HIP_CHECK(dev_malloc((void**)&c->d_best, 8));
        char* base = (char*)c->d_scratch;
        XformJob* dJ = (XformJob*)base;
        float* dT = (float*)(base + jb);
        float* dP = (float*)(base + jb + tb);
        int32_t* dO = (int32_t*)(base + jb + 2 * tb);          // n + 1 flags (the last one is 0)
        int32_t* dPos = (int32_t*)(base + jb + 2 * tb + ob);   // their exclusive scan; dPos[n] = accepted count
        int32_t* dB = (int32_t*)(base + jb + 2 * tb + 2 * ob);
        void* dTmp = base + jb + 2 * tb + 3 * ob;
        const unsigned int* d_unresolved = NULL;
        const int32_t* d_picks = NULL;
        if (device_picks) {
            char* pk = base + jb + 2 * tb + 3 * ob + scan_tmp;
            uint4* d_table = (uint4*)(pk + ((n * 16 + 255) / 256) * 256);
            d_picks = (const int32_t*)pk;
            const size_t lds = (size_t)(2 * (hmask + 1) + ((max_per_base + 1) & ~1)) * 4 + (size_t)max_per_base * 8;
            // The table goes up on the MAIN stream: a host-to-device copy of a few hundred kilobytes enqueued on the otherwise idle auxiliary
            // stream blocked the calling thread for 5.6 ms (64 trials of the ycb frame: 195 KB; measured, profiles/r04_trials_steps.json),
            // the same copy on the stream that carries the rest of the call's work returns at once.  The draws then run on the auxiliary
            // stream behind an event, next to the small bases' materialisation, which is enqueued behind the copy.
            STOCS_HIP_CHECK(hipMemcpyAsync(d_table, (char*)c->h_pin + PIN_VAR, batch ? 2 * tab_bytes : 16 * nbases, hipMemcpyHostToDevice, c->stream));
            STOCS_HIP_CHECK(hipEventRecord(c->ev_fork, c->stream));
            STOCS_HIP_CHECK(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            c->audit.use(0, d_table, true, "pick table", "table upload"); c->audit.record(c->ev_fork, 0); c->audit.wait(1, c->ev_fork);
            if (!c->bases.empty()) { const int rc0 = stocs_internal_prepare_small(c, max_per_base); if (rc0) return rc0; }
            hipLaunchKernelGGL(draw_picks_kernel, dim3((unsigned)c->bases.size()), dim3(256), lds, c->aux_stream, (const uint4*)d_table,
                               batch ? (const uint4*)((char*)d_table + tab_bytes) : (const uint4*)NULL, seed, max_per_base, hmask, (int4*)pk, dB);
            STOCS_HIP_CHECK(hipGetLastError());
            STOCS_HIP_CHECK(hipEventRecord(c->ev_join, c->aux_stream));
            STOCS_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            c->audit.use(1, d_table, false, "pick table", "draw picks"); c->audit.use(1, pk, true, "picks", "draw picks"); c->audit.use(1, dB, true, "job bases", "draw picks");
            c->audit.record(c->ev_join, 1); c->audit.wait(0, c->ev_join);
        }
        c->timing[1].lap("enqueue pick table upload + small bases + draws (auxiliary stream)");
        rc = stocs_internal_make_jobs(c, device_picks ? NULL : picks.data(), d_picks, (int)n, dJ, &d_unresolved);
        if (rc) return rc;
        c->timing[1].lap("enqueue resolve");
        tick("resolve picks");
        if (!device_picks) STOCS_HIP_CHECK(hipMemcpyAsync(dB, job_base.data(), n * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(rigid_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_spos, c->d_mpos, dJ, (int)n,
                           c->centroid_scene, c->centroid_model, dT, dP, dO);
        STOCS_HIP_CHECK(hipGetLastError());
        STOCS_HIP_CHECK(exclusive_scan(dTmp, scan_tmp, (const uint32_t*)dO, (uint32_t*)dPos, n + 1, c->stream));
        hipLaunchKernelGGL(compact_candidates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float4*)dT, (const float4*)dP, dO, dPos, dB,
                           (int)n, (float4*)cand_T(c), (float4*)cand_P(c), cand_lcp(c), cand_base(c), c->d_best);
        c->best_is_zero = true;
        STOCS_HIP_CHECK(hipGetLastError());
        int32_t* rb = (int32_t*)((char*)c->h_pin + PIN_TRANSFORMS);   // pinned read-back slot: accepted count, unresolved picks
        rb[0] = 0; rb[1] = 0;
        int32_t* tr_pin = (int32_t*)((char*)c->h_pin + PIN_VAR + 2 * tab_bytes);   // batch: first job of every trial in, first candidate of every trial out
        if (batch) {
            int32_t* d_tr = (int32_t*)(base + jb + 2 * tb + 3 * ob + scan_tmp + pb);
            for (size_t t = 0; t < n_tr; ++t) tr_pin[t] = (int32_t)first_job[(size_t)c->trial_first_base[t]];
            STOCS_HIP_CHECK(hipMemcpyAsync(d_tr, tr_pin, 4 * n_tr, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(gather_i32_kernel, dim3((unsigned)((n_tr + 255) / 256)), dim3(256), 0, c->stream, (const int32_t*)dPos, (const int32_t*)d_tr, (int)n_tr, d_tr + n_tr);
            STOCS_HIP_CHECK(hipGetLastError());
            STOCS_HIP_CHECK(hipMemcpyAsync(tr_pin, d_tr + n_tr, 4 * n_tr, hipMemcpyDeviceToHost, c->stream));
        }
        STOCS_HIP_CHECK(hipMemcpyAsync(&rb[0], dPos + n, 4, hipMemcpyDeviceToHost, c->stream));
        if (d_unresolved) STOCS_HIP_CHECK(hipMemcpyAsync(&rb[1], d_unresolved, 4, hipMemcpyDeviceToHost, c->stream));
        c->audit.use(0, d_picks, false, "picks", "resolve picks"); c->audit.use(0, dB, false, "job bases", "compact candidates");
        c->timing[1].lap("enqueue transform/scan/compact + read-backs");
        STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));   // the one synchronisation point of this call
        c->audit.host_sync(0);
        if (!c->audit.violations.empty()) {
            set_error("stocs_make_transforms: %zu stream-ordering violation(s); first: %s", c->audit.violations.size(), c->audit.violations[0].c_str());
            c->audit.violations.clear();
            return STOCS_ERR_STATE;
        }
        c->timing[1].lap("wait for the device");
        const int32_t n_ok = rb[0];
        const unsigned int n_unresolved = (unsigned int)rb[1];
        if (n_unresolved) { set_error("stocs_make_transforms: %u picks could not be resolved (internal inconsistency)", n_unresolved); return STOCS_ERR_STATE; }
        c->n_cands = n_ok;
        c->cands_stale = n_ok > 0;
        if (batch) c->trial_cand_off.assign(tr_pin, tr_pin + n_tr);
        tick("transform+compact");
    }
    if (n_candidates) *n_candidates = c->n_cands;
    return STOCS_OK;
}

// host mirror of the device-resident candidates (T, pose, lcp, base index), downloaded when somebody asks
static int ensure_host_candidates(stocs_ctx* c) {
    if (!c->cands_stale) return STOCS_OK;
    const size_t n = (size_t)c->n_cands;
    std::vector<float> T(n * 16), P(n * 16), l(n);
    std::vector<int32_t> b(n);
    STOCS_HIP_CHECK(hipMemcpyAsync(T.data(), cand_T(c), n * 64, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(P.data(), cand_P(c), n * 64, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(l.data(), cand_lcp(c), n * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(b.data(), cand_base(c), n * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->cands.resize(n);
    for (size_t j = 0; j < n; ++j) {
        memcpy(c->cands[j].T, &T[j * 16], 64);
        memcpy(c->cands[j].pose, &P[j * 16], 64);
        c->cands[j].lcp = l[j];   // 0 until stocs_verify_all ("score is not computed at this time", stocs.cpp:935-936)
        c->cands[j].base_index = b[j];
    }
    c->cands_stale = false;
    return STOCS_OK;
}

int stocs_get_candidates(stocs_ctx* c, float* T16, float* pose16, float* lcp, int32_t* base_index, int cap, int* n) {
    if (!c || !n) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    *n = c->n_cands;
    if (!(T16 || pose16 || lcp || base_index)) return STOCS_OK;
    int rc = ensure_host_candidates(c);
    if (rc) return rc;
    for (int i = 0; i < *n && i < cap; ++i) {
        if (T16) memcpy(T16 + (size_t)i * 16, c->cands[i].T, 64);
        if (pose16) memcpy(pose16 + (size_t)i * 16, c->cands[i].pose, 64);
        if (lcp) lcp[i] = c->cands[i].lcp;
        if (base_index) base_index[i] = c->cands[i].base_index;
    }
    return (*n > cap) ? STOCS_ERR_CAPACITY : STOCS_OK;
}

int stocs_verify_all(stocs_ctx* c, float* best_lcp, int* best_idx, float* best_pose16) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    const int n = c->n_cands;
    c->best_lcp = 0; c->best_index = -1;
    float pose[16];
    memset(pose, 0, sizeof(pose));
    c->timing[2].begin();
    if (n > 0) {
        // scores and the arg-max stay on the device: one LCP launch over the resident transforms, then
        // compute_best_transform (stocs.cpp:987-998: strict > from 0 => first maximum wins, Q18) as an integer max
        if (!c->d_best) STOCS_HIP_CHECK(dev_malloc((void**)&c->d_best, 8));
        int rc = launch_lcp(c, cand_T(c), n, cand_lcp(c), NULL, NULL, c->d_best, 0);   // scores + the arg-max key in one launch
        if (rc) return rc;
        rc = ensure_scratch(c, 256);
        if (rc) return rc;
        float* d_out = (float*)c->d_scratch;   // the transform jobs of this trial are done with the scratch area
        hipLaunchKernelGGL(winner_pose_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long*)c->d_best, (const float*)cand_P(c), n, d_out);
        STOCS_HIP_CHECK(hipGetLastError());
        if ((rc = ensure_pinned(c, PIN_VAR))) return rc;
        float* out18 = (float*)((char*)c->h_pin + PIN_VERIFY);   // pinned read-back slot (18 floats)
        STOCS_HIP_CHECK(hipMemcpyAsync(out18, d_out, 18 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        c->timing[2].lap("enqueue score + arg-max + winner");
        STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->timing[2].lap("wait for the device");
        uint32_t lo, hi;
        memcpy(&lo, &out18[0], 4); memcpy(&hi, &out18[1], 4);
        const uint64_t key = ((uint64_t)hi << 32) | lo;
        c->cands_stale = true;   // the host mirror (if any) lacks the new scores
        if (key) {
            uint32_t id = 0;
            stocs_unpack_best(key, &c->best_lcp, &id);
            c->best_index = (int)id;
            memcpy(pose, &out18[2], 64);
        }
    }
    if (best_lcp) *best_lcp = c->best_lcp;
    if (best_idx) *best_idx = c->best_index;
    if (best_pose16) memcpy(best_pose16, pose, 64);
    return STOCS_OK;
}

}  // extern "C"
