// ppf_index.hip -- the model point-pair-feature index on the device.
// Replaces the O(|M|^2) loop of stocs::pre_process_model (reference src/stocs.cpp:63-78),
// rgbd::ppf_map_insert (reference src/rgbd.cpp:123-154) and the std::map `find` at the call sites
// stocs.cpp:403,438,487,780,784.
//
// The reference stores every ordered pair under up to 128 keys (2 distance x 4x4x4 angle offsets);
// at |M| = 5000 that would be 3.2e9 entries.  Equivalent form used here: each pair is stored ONCE under
// its own quantised key F (CSR over the dense key space, pairs sorted by (id1,id2) inside a bucket);
//   exists(K)  = bit K of a bitmap that has the 128 offset keys of every non-empty bucket set;
//   lookup(K)  = union of the <=128 buckets F = K - o, merged in (id1,id2) order,
// which is exactly the reference's map content and insertion order (id1 outer, id2 inner loop).
#include <stdio.h>
#include <string.h>
#include <cstring>


#include <algorithm>

#include "prims.h"
#include "stocs_ctx.h"

namespace stocs {

// one thread per ordered pair (id1 = blockIdx.y*? ...): grid-stride over id1*M + id2
__global__ __launch_bounds__(256) void ppf_pair_keys_kernel(const float4* __restrict__ pos, const float4* __restrict__ nrm, int M,
                                                            int tr, int rot, int NA, int nD, uint64_t* __restrict__ keys,
                                                            uint32_t* __restrict__ hist) {
    const int64_t total = (int64_t)M * M;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int id1 = (int)(e / M), id2 = (int)(e % M);
        uint64_t out = ~0ull;  // id1 == id2 and out-of-range features sort to the end
        if (id1 != id2) {
            const float4 p1 = pos[id1], n1 = nrm[id1], p2 = pos[id2], n2 = nrm[id2];
            int f[4];
            ppf_compute(mk3(p1.x, p1.y, p1.z), mk3(n1.x, n1.y, n1.z), mk3(p2.x, p2.y, p2.z), mk3(n2.x, n2.y, n2.z), tr, rot, f);
            const int d = f[0] / tr, a1 = f[1] / rot, a2 = f[2] / rot, a3 = f[3] / rot;
            if (f[0] >= 0 && d < nD && f[1] >= 0 && a1 < NA && f[2] >= 0 && a2 < NA && f[3] >= 0 && a3 < NA) {
                const uint32_t key = ppf_pack(d, a1, a2, a3, NA);
                out = ((uint64_t)key << 32) | ((uint64_t)id1 << 16) | (uint64_t)id2;
                atomicAdd(&hist[key], 1u);
            }
        }
        keys[e] = out;
    }
}

__global__ __launch_bounds__(256) void ppf_unpack_pairs_kernel(const uint64_t* __restrict__ sorted, int64_t n, uint32_t* __restrict__ pairs) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) pairs[e] = (uint32_t)(sorted[e] & 0xFFFFFFFFull);
}

// set the 128 map keys of every non-empty bucket (rgbd.cpp:130-137)
__global__ __launch_bounds__(128) void ppf_exists_kernel(const uint32_t* __restrict__ hist, int64_t n_keys, int tr, int rot, int NA,
                                                         int nD, uint32_t* __restrict__ bits) {
    const int64_t key = blockIdx.x;
    if (key >= n_keys || hist[key] == 0) return;
    int rest = (int)key;
    const int a3 = rest % NA; rest /= NA;
    const int a2 = rest % NA; rest /= NA;
    const int a1 = rest % NA; rest /= NA;
    const int d = rest;
    const int o = threadIdx.x;  // 0..127
    const int o0 = (o >> 6) & 1, o1 = (o >> 4) & 3, o2 = (o >> 2) & 3, o3 = o & 3;
    // p1 in {F0 - tr, F0}; pk in {Fk - 2rot, Fk - rot, Fk, Fk + rot}
    const int K0 = d * tr - tr + o0 * tr;
    const int K1 = a1 * rot - 2 * rot + o1 * rot;
    const int K2 = a2 * rot - 2 * rot + o2 * rot;
    const int K3 = a3 * rot - 2 * rot + o3 * rot;
    if (K0 <= 5 || K1 < 0 || K2 < 0 || K3 < 0) return;  // rgbd.cpp:136 (literal 5)
    const int kd = K0 / tr, k1 = K1 / rot, k2 = K2 / rot, k3 = K3 / rot;
    if (kd >= nD || k1 >= NA || k2 >= NA || k3 >= NA) return;  // never produced by ppf_compute
    const uint32_t K = ppf_pack(kd, k1, k2, k3, NA);
    atomicOr(&bits[K >> 5], 1u << (K & 31));
}

int build_ppf_index(stocs_ctx* c) {
    PpfIndex& ix = c->index;
    const int M = c->nM;
    ix.tr = c->prm.ppf_tr_discretization;
    ix.rot = c->prm.ppf_rot_discretization;
    ix.NA = 180 / ix.rot + 1;
    // distance bins: upper bound from the bounding-box diagonal (+ rounding up of closest_bin)
    {
        V3 mn = mk3(1e30f, 1e30f, 1e30f), mx = mk3(-1e30f, -1e30f, -1e30f);
        for (int i = 0; i < M; ++i) {
            const V3 p = c->h_mpos_raw[i];
            mn = mk3(std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z));
            mx = mk3(std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z));
        }
        const double diag = M ? sqrt((double)sqn3(mx - mn)) : 0.0;
        ix.nD = (int)(diag * 1000.0) / ix.tr + 3;
    }
    ix.n_keys = (int64_t)ix.nD * ix.NA * ix.NA * ix.NA;
    const int64_t total = (int64_t)M * M;
    if (ix.n_keys > (int64_t)1 << 31 || total > (int64_t)400 * 1000 * 1000) {
        set_error("PPF index too large (|M|=%d, key space %lld): build_index is meant for |M| <= 20000", M, (long long)ix.n_keys);
        return STOCS_ERR_INVALID;
    }
    uint32_t* d_hist = NULL;
    uint64_t *d_keys = NULL, *d_sorted = NULL;
    void* d_tmp = NULL;
    size_t tmp_bytes = 0;
    // the temporaries are released on every exit path (the HIP checks below return early)
    struct TmpGuard {
        uint32_t** a; uint64_t** b; uint64_t** c; void** d;
        ~TmpGuard() { if (*a) (void)hipFree(*a); if (*b) (void)hipFree(*b); if (*c) (void)hipFree(*c); if (*d) (void)hipFree(*d); }
    } tmp_guard = {&d_hist, &d_keys, &d_sorted, &d_tmp};
    const size_t words = (size_t)((ix.n_keys + 31) / 32);
    STOCS_HIP_CHECK(dev_malloc((void**)&d_hist, (size_t)(ix.n_keys + 1) * 4));
    STOCS_HIP_CHECK(hipMemsetAsync(d_hist, 0, (size_t)(ix.n_keys + 1) * 4, c->stream));
    STOCS_HIP_CHECK(dev_malloc((void**)&d_keys, (size_t)std::max<int64_t>(total, 1) * 8));
    STOCS_HIP_CHECK(dev_malloc((void**)&d_sorted, (size_t)std::max<int64_t>(total, 1) * 8));
    STOCS_HIP_CHECK(dev_malloc((void**)&ix.d_exists, std::max<size_t>(words, 1) * 4));
    STOCS_HIP_CHECK(hipMemsetAsync(ix.d_exists, 0, std::max<size_t>(words, 1) * 4, c->stream));
    if (total > 0) {
        const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 64);
        hipLaunchKernelGGL(ppf_pair_keys_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_mpos_raw, c->d_mnrm, M, ix.tr, ix.rot,
                           ix.NA, ix.nD, d_keys, d_hist);
        STOCS_HIP_CHECK(hipGetLastError());
        int key_bits = 1;
        while (((int64_t)1 << key_bits) < ix.n_keys) key_bits++;
        const unsigned end_bit = 64;  // invalid entries (all ones) must sort last
        (void)key_bits;
        STOCS_HIP_CHECK(sort_keys(NULL, tmp_bytes, d_keys, d_sorted, (size_t)total, 0, end_bit, c->stream));
        STOCS_HIP_CHECK(dev_malloc(&d_tmp, tmp_bytes));
        STOCS_HIP_CHECK(sort_keys(d_tmp, tmp_bytes, d_keys, d_sorted, (size_t)total, 0, end_bit, c->stream));
        hipLaunchKernelGGL(ppf_exists_kernel, dim3((unsigned)ix.n_keys), dim3(128), 0, c->stream, d_hist, ix.n_keys, ix.tr, ix.rot,
                           ix.NA, ix.nD, ix.d_exists);
        STOCS_HIP_CHECK(hipGetLastError());
    }
    // bucket offsets: host scan of the histogram (one-time; the host copy also plans the lookups)
    std::vector<uint32_t> hist((size_t)ix.n_keys + 1, 0);
    STOCS_HIP_CHECK(hipMemcpyAsync(hist.data(), d_hist, (size_t)ix.n_keys * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    ix.h_bucket_start.assign((size_t)ix.n_keys + 1, 0);
    uint64_t run = 0;
    ix.n_nonempty_buckets = 0;
    for (int64_t k = 0; k < ix.n_keys; ++k) {
        ix.h_bucket_start[(size_t)k] = (uint32_t)run;
        run += hist[(size_t)k];
        ix.n_nonempty_buckets += hist[(size_t)k] != 0;
    }
    ix.h_bucket_start[(size_t)ix.n_keys] = (uint32_t)run;
    ix.n_pairs = (int64_t)run;
    STOCS_HIP_CHECK(dev_malloc((void**)&ix.d_bucket_start, (size_t)(ix.n_keys + 1) * 4));
    STOCS_HIP_CHECK(hipMemcpyAsync(ix.d_bucket_start, ix.h_bucket_start.data(), (size_t)(ix.n_keys + 1) * 4, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(dev_malloc((void**)&ix.d_pairs, (size_t)std::max<int64_t>(ix.n_pairs, 1) * 4));
    if (ix.n_pairs > 0) {
        hipLaunchKernelGGL(ppf_unpack_pairs_kernel, dim3((unsigned)((ix.n_pairs + 255) / 256)), dim3(256), 0, c->stream, d_sorted,
                           ix.n_pairs, ix.d_pairs);
        STOCS_HIP_CHECK(hipGetLastError());
    }
    // host copy of the existence bitmap for the single-key query API
    ix.h_exists.assign(std::max<size_t>(words, 1), 0);
    STOCS_HIP_CHECK(hipMemcpyAsync(ix.h_exists.data(), ix.d_exists, std::max<size_t>(words, 1) * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    ix.n_exist_keys = 0;
    for (size_t w = 0; w < ix.h_exists.size(); ++w) ix.n_exist_keys += __builtin_popcount(ix.h_exists[w]);
    ix.built = true;
    return STOCS_OK;
}

// host-side planning shared with congruent.hip: the source buckets of lookup(K) as CSR ranges in ascending index position
int plan_lookup(const PpfIndex& ix, const int* K, std::vector<std::pair<uint32_t, uint32_t> >* ranges) {
    ranges->clear();
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return 0;
    if (K[0] % ix.tr || K[1] % ix.rot || K[2] % ix.rot || K[3] % ix.rot) return 0;
    int64_t total = 0;
    // F = K - o, o0 in {-tr,0}, ok in {-2rot,-rot,0,rot}, visited in ASCENDING key order (every F component ascending), so
    // the ranges come out in ascending index position; adjacent buckets are merged on the fly (the 4 consecutive a3 bins of
    // a lookup are neighbours in the CSR array).  The gathered list of a lookup is then in INDEX ORDER -- ascending
    // quantised feature, then ascending (id1, id2) -- which is what the enumeration of congruent.hip is defined on.
    for (int a = 0; a < 2; ++a)
        for (int b = 3; b >= 0; --b)
            for (int cc = 3; cc >= 0; --cc)
                for (int d = 3; d >= 0; --d) {
                    const int F0 = K[0] + a * ix.tr, F1 = K[1] + (2 - b) * ix.rot, F2 = K[2] + (2 - cc) * ix.rot, F3 = K[3] + (2 - d) * ix.rot;
                    if (F1 < 0 || F2 < 0 || F3 < 0) continue;
                    const int fd = F0 / ix.tr, f1 = F1 / ix.rot, f2 = F2 / ix.rot, f3 = F3 / ix.rot;
                    if (fd >= ix.nD || f1 >= ix.NA || f2 >= ix.NA || f3 >= ix.NA) continue;
                    const uint32_t key = ppf_pack(fd, f1, f2, f3, ix.NA);
                    const uint32_t s = ix.h_bucket_start[key], e = ix.h_bucket_start[key + 1];
                    if (e <= s) continue;
                    total += e - s;
                    if (!ranges->empty() && ranges->back().second == s) ranges->back().second = e;
                    else ranges->push_back(std::make_pair(s, e));
                }
    return (int)std::min<int64_t>(total, 0x7fffffff);
}

// touches the lines of the bucket table that plan_lookup(K) will read (the 4 a3 bins of a probe share a line or two)
void prefetch_lookup(const PpfIndex& ix, const int* K) {
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return;
    if (K[0] % ix.tr || K[1] % ix.rot || K[2] % ix.rot || K[3] % ix.rot) return;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b)
            for (int cc = 0; cc < 4; ++cc) {
                const int F0 = K[0] + a * ix.tr, F1 = K[1] + (2 - b) * ix.rot, F2 = K[2] + (2 - cc) * ix.rot, F3 = K[3] - ix.rot;
                if (F1 < 0 || F2 < 0) continue;
                const int fd = F0 / ix.tr, f1 = F1 / ix.rot, f2 = F2 / ix.rot, f3 = F3 < 0 ? 0 : F3 / ix.rot;
                if (fd >= ix.nD || f1 >= ix.NA || f2 >= ix.NA || f3 >= ix.NA) continue;
                const uint32_t* p = &ix.h_bucket_start[ppf_pack(fd, f1, f2, f3, ix.NA)];
                __builtin_prefetch(p);
                __builtin_prefetch(p + 4);
            }
}

}  // namespace stocs

using namespace stocs;

extern "C" {

int stocs_ppf_compute_host(const float* p1, const float* n1, const float* p2, const float* n2, int tr, int rot, int32_t* key4) {
    if (!p1 || !n1 || !p2 || !n2 || !key4 || tr <= 0 || rot <= 0) return STOCS_ERR_INVALID;
    int f[4];
    ppf_compute(mk3(p1[0], p1[1], p1[2]), mk3(n1[0], n1[1], n1[2]), mk3(p2[0], p2[1], p2[2]), mk3(n2[0], n2[1], n2[2]), tr, rot, f);
    for (int k = 0; k < 4; ++k) key4[k] = f[k];
    return STOCS_OK;
}

int stocs_index_exists(const stocs_ctx* c, const int32_t* K, int* exists) {
    if (!c || !K || !exists) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->index.built) { set_error("PPF index not built"); return STOCS_ERR_STATE; }
    const PpfIndex& ix = c->index;
    *exists = 0;
    if (K[0] <= 5 || K[1] < 0 || K[2] < 0 || K[3] < 0) return STOCS_OK;
    if (K[0] % ix.tr || K[1] % ix.rot || K[2] % ix.rot || K[3] % ix.rot) return STOCS_OK;
    const int kd = K[0] / ix.tr, k1 = K[1] / ix.rot, k2 = K[2] / ix.rot, k3 = K[3] / ix.rot;
    if (kd >= ix.nD || k1 >= ix.NA || k2 >= ix.NA || k3 >= ix.NA) return STOCS_OK;
    const uint32_t key = ppf_pack(kd, k1, k2, k3, ix.NA);
    *exists = (ix.h_exists[key >> 5] >> (key & 31)) & 1;
    return STOCS_OK;
}

int stocs_index_lookup(stocs_ctx* c, const int32_t* K, int32_t* pairs2, int64_t cap, int64_t* n) {
    if (!c || !K || !n) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->index.built) { set_error("PPF index not built"); return STOCS_ERR_STATE; }
    std::vector<std::pair<uint32_t, uint32_t> > ranges;
    int key[4] = {K[0], K[1], K[2], K[3]};
    *n = plan_lookup(c->index, key, &ranges);
    if (!pairs2 || cap <= 0) return STOCS_OK;
    std::vector<uint32_t> all((size_t)*n);
    size_t off = 0;
    for (size_t r = 0; r < ranges.size(); ++r) {
        const size_t len = ranges[r].second - ranges[r].first;
        STOCS_HIP_CHECK(hipMemcpyAsync(all.data() + off, c->index.d_pairs + ranges[r].first, len * 4, hipMemcpyDeviceToHost, c->stream));
        off += len;
    }
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::sort(all.begin(), all.end());  // (id1<<16|id2) ascending == lexicographic (id1,id2)
    for (int64_t i = 0; i < *n && i < cap; ++i) { pairs2[2 * i] = (int32_t)(all[(size_t)i] >> 16); pairs2[2 * i + 1] = (int32_t)(all[(size_t)i] & 0xFFFF); }
    return (*n > cap) ? STOCS_ERR_CAPACITY : STOCS_OK;
}

// ---- on-disk index (flat little-endian file; replaces the Boost archive of rgbd.cpp:156-177) ----
struct IndexFileHeader {
    char magic[8];       // "STOCSIX1"
    int32_t tr, rot, NA, nD, nM, reserved;
    int64_t n_keys, n_pairs;
    uint64_t model_hash; // FNV-1a over the raw model positions and the normalised normals
};
static uint64_t model_hash(const stocs_ctx* c) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    mix(c->h_mpos_raw.data(), c->h_mpos_raw.size() * sizeof(V3));
    mix(c->h_mnrm.data(), c->h_mnrm.size() * sizeof(V3));
    return h;
}

int stocs_index_save(stocs_ctx* c, const char* path) {
    if (!c || !path) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->index.built) { set_error("stocs_index_save: PPF index not built"); return STOCS_ERR_STATE; }
    const PpfIndex& ix = c->index;
    std::vector<uint32_t> pairs((size_t)std::max<int64_t>(ix.n_pairs, 1));
    STOCS_HIP_CHECK(hipMemcpyAsync(pairs.data(), ix.d_pairs, (size_t)ix.n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("stocs_index_save: cannot open %s", path); return STOCS_ERR_INVALID; }
    IndexFileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "STOCSIX1", 8);
    h.tr = ix.tr; h.rot = ix.rot; h.NA = ix.NA; h.nD = ix.nD; h.nM = c->nM; h.n_keys = ix.n_keys; h.n_pairs = ix.n_pairs;
    h.model_hash = model_hash(c);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(ix.h_bucket_start.data(), 4, (size_t)ix.n_keys + 1, f) == (size_t)ix.n_keys + 1 &&
              fwrite(pairs.data(), 4, (size_t)ix.n_pairs, f) == (size_t)ix.n_pairs &&
              fwrite(ix.h_exists.data(), 4, ix.h_exists.size(), f) == ix.h_exists.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("stocs_index_save: write to %s failed", path); return STOCS_ERR_INVALID; }
    return STOCS_OK;
}

int stocs_index_load(stocs_ctx* c, const char* path) {
    if (!c || !path) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (c->index.built) { set_error("stocs_index_load: the context already has an index"); return STOCS_ERR_STATE; }
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("stocs_index_load: cannot open %s", path); return STOCS_ERR_INVALID; }
    IndexFileHeader h;
    PpfIndex& ix = c->index;
    int rc = STOCS_OK;
    std::vector<uint32_t> pairs;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "STOCSIX1", 8) != 0) { set_error("stocs_index_load: %s is not a STOCSIX1 file", path); rc = STOCS_ERR_INVALID; }
    else if (h.nM != c->nM || h.tr != c->prm.ppf_tr_discretization || h.rot != c->prm.ppf_rot_discretization || h.model_hash != model_hash(c)) {
        set_error("stocs_index_load: %s was built for another model cloud or discretisation", path); rc = STOCS_ERR_INVALID;
    } else if (h.n_keys <= 0 || h.n_keys > ((int64_t)1 << 31) || h.n_pairs < 0 || h.n_pairs > (int64_t)400 * 1000 * 1000 || h.NA != 180 / h.rot + 1 ||
               h.n_keys != (int64_t)h.nD * h.NA * h.NA * h.NA) {
        set_error("stocs_index_load: corrupt header in %s", path); rc = STOCS_ERR_INVALID;
    } else {
        const size_t words = (size_t)((h.n_keys + 31) / 32);
        ix.h_bucket_start.assign((size_t)h.n_keys + 1, 0);
        pairs.resize((size_t)std::max<int64_t>(h.n_pairs, 1));
        ix.h_exists.assign(std::max<size_t>(words, 1), 0);
        if (fread(ix.h_bucket_start.data(), 4, (size_t)h.n_keys + 1, f) != (size_t)h.n_keys + 1 || fread(pairs.data(), 4, (size_t)h.n_pairs, f) != (size_t)h.n_pairs ||
            fread(ix.h_exists.data(), 4, words, f) != words || ix.h_bucket_start[(size_t)h.n_keys] != (uint32_t)h.n_pairs) {
            set_error("stocs_index_load: truncated or inconsistent file %s", path); rc = STOCS_ERR_INVALID;
        }
    }
    fclose(f);
    if (!rc) {
        // one O(n_keys + n_pairs) pass: a file that passes is safe to index with (offsets monotone and in range, every
        // model id below nM, no existence bit beyond the key space)
        const char* why = NULL;
        for (int64_t k = 0; k < h.n_keys && !why; ++k)
            if (ix.h_bucket_start[(size_t)k] > ix.h_bucket_start[(size_t)k + 1]) why = "bucket offsets are not monotone";
        if (!why && ix.h_bucket_start[0] != 0) why = "first bucket offset is not 0";
        const uint32_t nM = (uint32_t)c->nM;
        for (int64_t e = 0; e < h.n_pairs && !why; ++e)
            if ((pairs[(size_t)e] >> 16) >= nM || (pairs[(size_t)e] & 0xFFFFu) >= nM) why = "pair id outside the model";
        if (!why && (h.n_keys & 31)) {
            const uint32_t tail = ix.h_exists.back() >> (h.n_keys & 31);
            if (tail) why = "existence bits beyond the key space";
        }
        if (why) { set_error("stocs_index_load: inconsistent file %s (%s)", path, why); rc = STOCS_ERR_INVALID; }
    }
    if (rc) { ix.h_bucket_start.clear(); ix.h_exists.clear(); return rc; }
    ix.tr = h.tr; ix.rot = h.rot; ix.NA = h.NA; ix.nD = h.nD; ix.n_keys = h.n_keys; ix.n_pairs = h.n_pairs;
    ix.n_nonempty_buckets = 0;
    for (int64_t k = 0; k < ix.n_keys; ++k) ix.n_nonempty_buckets += ix.h_bucket_start[(size_t)k + 1] != ix.h_bucket_start[(size_t)k];
    ix.n_exist_keys = 0;
    for (size_t w = 0; w < ix.h_exists.size(); ++w) ix.n_exist_keys += __builtin_popcount(ix.h_exists[w]);
    STOCS_HIP_CHECK(dev_malloc((void**)&ix.d_bucket_start, (size_t)(ix.n_keys + 1) * 4));
    STOCS_HIP_CHECK(dev_malloc((void**)&ix.d_pairs, (size_t)std::max<int64_t>(ix.n_pairs, 1) * 4));
    STOCS_HIP_CHECK(dev_malloc((void**)&ix.d_exists, ix.h_exists.size() * 4));
    STOCS_HIP_CHECK(hipMemcpyAsync(ix.d_bucket_start, ix.h_bucket_start.data(), (size_t)(ix.n_keys + 1) * 4, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(ix.d_pairs, pairs.data(), (size_t)ix.n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipMemcpyAsync(ix.d_exists, ix.h_exists.data(), ix.h_exists.size() * 4, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    ix.built = true;
    return STOCS_OK;
}

int stocs_index_stats(const stocs_ctx* c, int64_t* n_pairs, int64_t* n_buckets, int64_t* n_keys) {
    if (!c) return STOCS_ERR_INVALID;
    if (!c->index.built) { set_error("PPF index not built"); return STOCS_ERR_STATE; }
    if (n_pairs) *n_pairs = c->index.n_pairs;
    if (n_buckets) *n_buckets = c->index.n_nonempty_buckets;
    if (n_keys) *n_keys = c->index.n_exist_keys;
    return STOCS_OK;
}

}  // extern "C"
