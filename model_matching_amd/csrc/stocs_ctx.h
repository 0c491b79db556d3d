// stocs_ctx.h -- internal state behind the opaque stocs_ctx of include/stocs_hip.h.
#ifndef STOCS_CTX_H
#define STOCS_CTX_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>

#include <algorithm>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/stocs_hip.h"
#include "stocs_math.h"
#include "stream_audit.h"

namespace stocs {

void set_error(const char* fmt, ...);

#define STOCS_HIP_CHECK(expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            stocs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return STOCS_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

// Every device allocation of the library goes through here and is counted (stocs_device_alloc_count): a warm context
// must run trial after trial without allocating, and the tests assert exactly that.
extern unsigned long long g_dev_allocs;
inline hipError_t dev_malloc(void** p, size_t bytes) {
    __atomic_fetch_add(&g_dev_allocs, 1ull, __ATOMIC_RELAXED);
    return hipMalloc(p, bytes);
}
template <class T> inline hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc((void**)p, bytes); }
// pinned host memory goes through the same counter: a hipHostMalloc / hipHostFree inside a trial is as much of a stall
// as a hipMalloc, and stocs_device_alloc_count must see it
inline hipError_t pinned_malloc(void** p, size_t bytes) {
    __atomic_fetch_add(&g_dev_allocs, 1ull, __ATOMIC_RELAXED);
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

// Host wall clock of the steps of one entry point, always recorded (a handful of clock reads, no synchronisation of its
// own): when a call takes 80 ms instead of 1, the record says which step it spent them in (stocs_last_call_timing).
struct CallTiming {
    enum { MAX_STEPS = 18 };
    int n;
    const char* label[MAX_STEPS];
    double ms[MAX_STEPS];
    double t_last;
    static double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    void begin() { n = 0; t_last = now_s(); }
    void lap(const char* what) {
        const double t = now_s();
        if (n < MAX_STEPS) { label[n] = what; ms[n] = (t - t_last) * 1e3; ++n; }
        t_last = t;
    }
};

// Binds the calling thread to the context's device for the duration of an entry point and restores the
// caller's device afterwards (a context may be driven from any thread, one thread at a time).
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Grow-only device workspace: slabs of plain hipMalloc memory handed out by bumping an offset.  reset() recycles
// everything (nothing of it may still be in flight); a slab that turned out too small is joined by a bigger one
// and the slabs are merged at the next reset, so a steady-state caller never calls hipMalloc / hipFree.
struct Slab { char* p; size_t cap, used; };
struct Arena {
    std::vector<Slab> slabs;
    // nothing that lives in the arena may still be in flight on the device
    int reset() {
        if (slabs.size() > 1) {
            size_t tot = 0;
            for (size_t i = 0; i < slabs.size(); ++i) { tot += slabs[i].cap; (void)hipFree(slabs[i].p); }
            slabs.clear();
            tot += tot;       // slack: trials of one scene differ in size by tens of percent, and 288 GB of HBM make room cheap
            Slab sl = {NULL, tot, 0};
            STOCS_HIP_CHECK(dev_malloc((void**)&sl.p, tot));
            slabs.push_back(sl);
        }
        for (size_t i = 0; i < slabs.size(); ++i) slabs[i].used = 0;
        return STOCS_OK;
    }
    // right after reset(): make sure ONE slab can hold `bytes` (a good estimate up front avoids growing in pieces)
    // headroom 2: a later trial of the same scene with more pair-list entries must not regrow the slab (a 0.6 GB
    // hipFree + hipMalloc inside a trial was the 78 ms outlier of BENCH_r01's pipeline run 4).  A piece of a trial batch is sized
    // against a memory ceiling and asks for little more than it needs; when the device cannot give that either the caller gets
    // STOCS_ERR_NOMEM and cuts the piece (stocs_run_trials).
    int reserve(size_t bytes, double headroom = 2.0) {
        if (slabs.size() == 1 && slabs[0].cap >= bytes) return STOCS_OK;
        destroy();
        Slab sl = {NULL, (size_t)((double)bytes * headroom), 0};
        const hipError_t e = dev_malloc((void**)&sl.p, sl.cap);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); stocs::set_error("arena of %zu bytes: out of device memory", sl.cap); return STOCS_ERR_NOMEM; }
        STOCS_HIP_CHECK(e);
        slabs.push_back(sl);
        return STOCS_OK;
    }
    int take(size_t bytes, void** out) {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        for (size_t i = 0; i < slabs.size(); ++i)
            if (slabs[i].cap - slabs[i].used >= bytes) { *out = slabs[i].p + slabs[i].used; slabs[i].used += bytes; return STOCS_OK; }
        Slab sl = {NULL, std::max<size_t>(bytes + bytes / 4, slabs.empty() ? ((size_t)64 << 20) : 2 * slabs.back().cap), 0};
        STOCS_HIP_CHECK(dev_malloc((void**)&sl.p, sl.cap));
        sl.used = bytes;
        slabs.push_back(sl);
        *out = sl.p;
        return STOCS_OK;
    }
    void destroy() { for (size_t i = 0; i < slabs.size(); ++i) (void)hipFree(slabs[i].p); slabs.clear(); }
};

// Brick grid over the centred scene (replaces the kd-tree of kdtree.h for the restricted-radius
// nearest-neighbour query).  Cell edge h = epsilon.  A cell's candidate list holds every scene
// point whose distance to the cell's box is <= epsilon (+ a 0.1% safety margin), so ONE list scan
// answers the query exactly.  Bricks of 8x8x8 cells exist only where some cell is non-empty.
struct SceneGrid {
    float ox, oy, oz, inv_h;
    int nx, ny, nz;      // cells per axis
    int nbx, nby, nbz;   // bricks per axis
    int n_bricks;
    int64_t n_entries;
    double avg_list_len;   // entries per non-empty cell (of the lists as stored: after the dominance pruning, when it is on)
    double avg_dilated_len;   // ... of every scene point within r of the cell box (what the list held before pruning)
    bool pruned;           // the lists hold only points that can be the nearest neighbour of some position in their cell (grid.hip)
    int32_t* d_top;      // nbx*nby*nbz -> brick id or -1
    uint4* d_flat;       // the same cell words addressed directly, (cz*ny + cy)*nx + cx, when the box is small enough (sparse
                         // scenes, cell edge eps): saves the LCP kernel the dependent `top` look-up; else NULL
    uint4* d_cells;      // n_bricks*512: (offset, count, sub-cell mask lo, hi); mask bit s set <=> some scene
                         // point lies within epsilon of sub-cell s (4x4x4 sub-cells, x fastest)
    float4* d_list;      // (x, y, z, bits(scene index))
    float* d_chunk_r;    // dense scenes only: lists sorted by distance from the cell centre; per 8-entry chunk a
                         // lower bound of that distance (early exit by the triangle inequality); else NULL
    float h;             // cell edge
    bool has_nearest;    // dense grids with cell edges below epsilon: the z word of a cell is a lower bound of the distance from the
                         // cell centre to the nearest listed point (the sub-cell mask it replaces is all ones there)
    // Distance field of the scene on a coarse grid (cell edge cg_g >= epsilon), for the patch test of the scan kernels (lcp.hip):
    // d_dist[cell] = distance from the cell's centre to the nearest scene point, rounded down, capped at cg_cap.  The box covers the
    // scene's bounding box widened by cg_cap + cg_g on every side, so a position outside it is at least cg_cap from every scene point.
    // The memory is taken with the grid (no allocation later); the values are filled on first use (fill_cull_field).
    float* d_dist;
    bool dist_ready;
    float cg_ox, cg_oy, cg_oz, cg_g, cg_inv_g, cg_cap;
    int cg_nx, cg_ny, cg_nz, cg_w;   // cg_w: cells a point reaches on either side (ceil(cap / g))
    double bb_mn[3], bb_mx[3];       // bounding box of the centred scene
};

// Model PPF index on the device: every ordered pair stored once under its own quantised key F
// (CSR over the dense key space), plus the dilated existence bitmap ("is K a key of the
// reference's 128-fold map?").
struct PpfIndex {
    bool built;
    int tr, rot, NA, nD;          // angle bins NA = 180/rot + 1, distance bins nD
    int64_t n_keys;               // nD*NA^3
    int64_t n_pairs;
    uint32_t* d_bucket_start;     // n_keys + 1
    uint32_t* d_pairs;            // n_pairs: (id1 << 16) | id2, sorted inside each bucket
    uint32_t* d_exists;           // bitmap over the key space, n_keys bits
    std::vector<uint32_t> h_bucket_start;  // host copy (planning of the lookups)
    std::vector<uint32_t> h_exists;        // host copy of the bitmap (single-key query API)
    int64_t n_nonempty_buckets, n_exist_keys;
};

struct BaseRec {
    int ids[4];
    float inv1, inv2;
};

// rows 6-7 on the device: ordered base + invariants of one attempt (try_sampled_base, stocs.cpp:224-268)
struct BaseOut { int32_t ids[4]; float inv[2]; int32_t valid; int32_t pad; };

struct Candidate {
    float T[16];     // centred frames (scored), stocs.cpp:923
    float pose[16];  // camera frame (returned), stocs.cpp:925-937
    float lcp;
    int base_index;
};

struct Thresholds {
    float lcp_dot_lo;   // normal test true  <=>  lcp_dot_lo <= dot <= 1
    float ang_dot_hi;   // internal-angle reject <=> (ang_dot_hi <= dot <= 1) || (-1 <= dot <= ang_dot_lo)
    float ang_dot_lo;
};

}  // namespace stocs

struct stocs_ctx {
    stocs_params prm;
    int device;
    hipStream_t stream;       // the stream all work is issued on
    hipStream_t own_stream;   // created with the context; `stream` may point to a caller's stream instead
    hipStream_t aux_stream;   // second stream of the context for work that is independent of `stream` until an event joins it
    hipEvent_t ev0, ev1, ev_fork, ev_join;
    hipEvent_t ev_t[10];   // timing events around the device groups of stocs_find_congruent_all (always recorded; read after the call's own sync)
    int nS, nM;
    stocs::Thresholds thr;

    // host copies (centred unless stated)
    std::vector<stocs::V3> h_spos, h_snrm, h_mpos, h_mnrm, h_mpos_raw, h_munit;
    std::vector<float> h_sprob;       // class_probability_ (decays in instance mode)
    std::vector<float> h_sprob0;      // class probabilities as given at construction (stocs_reset_trial)
    std::vector<int32_t> h_spix;      // row, col
    stocs::V3 centroid_scene, centroid_model, gcenter;
    float ratio;
    std::vector<int32_t> h_mperm;     // patch order of the model used by the LCP kernel (sorted slot -> model index)

    // device clouds
    char* d_scene_mem; size_t scene_cap;   // one grow-only slab for the three scene arrays below (a new frame reuses it)
    float4* d_spos;    // xyz + class prob
    float4* d_snrmw;   // unit normal + class prob weight (what LCP adds, stocs.cpp:1033)
    int2* d_spix;
    float4* d_mpos;    // centred model, original order (xyz, 0)
    float4* d_mnrm;
    float4* d_munit;   // unit-cube model (pairCreationFunctor.h:96-132)
    float4* d_mpos_raw;  // un-shifted model positions (index build)
    float4* d_mpos_s;  // copies in patch order for the LCP kernel (positions NaN-padded to whole 64-point steps + one)
    float4* d_mnrm_s;
    int32_t* d_mperm;
    float4* d_mpatch;  // per 64-point step of the sorted model: bounding sphere (centre xyz, radius) of its points (patch test, lcp.hip)
    float patch_r_ref; // the radius most patches stay below (sizes the cap of the scene's distance field)
    int lcp_cull;      // 1: the scan kernels skip the 64-point steps whose bounding sphere is farther than epsilon from every scene point
    int lcp_group;     // lanes per queued query in the verify trips: 4 (default, two list entries per lane) or 8 (one entry per lane)
    double lcp_cull_after;   // lcp_cull == 1: the distance field is filled once this many point queries were scored against the scene (default 1e9)
    bool prev_scene_warm;    // the scene before this one crossed that threshold: a stream of frames will again, so the field of a new frame
                             // is filled right away, on the auxiliary stream (cull_pending: the scoring stream has not waited for it yet)
    bool cull_pending;
    hipEvent_t ev_cull;
    int scene_scored;  // scoring launches against the current scene
    double scene_work; // candidates x model points scored against the current scene so far (the distance field is filled when it pays)

    stocs::SceneGrid grid;
    stocs::Arena grid_mem;   // top / cells / list / chunk_r of the current grid (reset by every build)
    stocs::Arena grid_ws;    // temporaries of a grid build
    int grid_div;   // cell edge = epsilon / grid_div
    int grid_prune; // 1 (default): the grid's lists are dominance-pruned (grid.hip); 0: the layouts of rounds 2-4 (STOCS_GRID_PRUNE, read at stocs_ctx_create)
    int lcp_variant;   // -1: STOCS_LCP_VARIANT or automatic; else stocs_set_option("lcp_variant")
    int device_clock;  // 1: stocs_find_congruent_all records HIP events between its kernel groups ("device: ..." steps of stocs_last_call_timing); default 0 (STOCS_DEVICE_CLOCK=1 turns it on)
    int lcp_split;     // 1: four wavefronts share one candidate (default), 0: one wavefront per candidate
    int lcp_flat;      // 1: build and use the flat cell table when it fits (default), 0: brick look-ups only
    int lcp_order;     // 0: candidates in batch order; 1: spatially ordered processing of big batches; 2: + XCD-contiguous blocks
    void* d_order;     // keys / permutation / sort scratch of the ordering
    size_t order_bytes;
    // class-mode sampling, lean kernel (sample.hip): exclusive prefix sums of the prior's 2^32 fixed-point weights in scene order (S + 1
    // entries), recomputed when the class probabilities on the device have changed (prior_epoch) or the scene has (cdf_n)
    void* d_cdf; size_t cdf_bytes, cdf_n; unsigned long long prior_epoch, cdf_epoch;
    stocs::PpfIndex index;

    // image-space state of instance mode (stocs.hpp:153-155)
    bool has_edge;
    std::vector<uint8_t> edge_map;   // png values (all zero = "file absent", stocs.cpp:117-119)
    void* inst;                      // device-side state of instance mode (sample.hip, InstanceState): runs of the edge map,
                                     // previous_segment / segmentation_buffer / seg_mask_<n> per scene point, the decaying prior

    std::vector<int32_t> last_segment;   // `segment` of the last instance-mode attempt (stocs.cpp:628-638)

    // run state
    std::vector<stocs::BaseRec> bases;
    // a batch of independent trials in one set of launches (stocs_run_trials, trials.hip): `bases` is then the concatenation of the
    // trials' base sets, and these say whose each base is.  All empty outside such a batch (one trial, the calls' own seed).
    std::vector<uint64_t> base_seed;        // per base: seed of its trial
    std::vector<int32_t> base_local;        // per base: its slot in its own trial's base set
    std::vector<int32_t> trial_first_base;  // per trial of the batch (+ 1): first base
    std::vector<int32_t> trial_cand_off;    // per trial of the batch (+ 1): first candidate (filled by stocs_make_transforms)
    const float4* snrmw_trial0;             // instance-mode batches: trial t scores against the weights at snrmw_trial0 + t * snrmw_stride bytes
    size_t snrmw_stride;
    const int32_t* lcp_cand_trial;          // != NULL (with snrmw_override = trial 0's copy): candidate i of the launch scores against trial lcp_cand_trial[i]'s copy
    const float4* snrmw_override;           // != NULL: the scoring kernel reads its scene normals + weights here (one trial of such a batch)
    void* trials;                           // results of the last stocs_run_trials (trials.hip, TrialBatch)
    // congruent quads: only their per-base counts live here (quad_off[b+1] - quad_off[b]); `cong` (congruent.hip,
    // CongruentState) keeps what is needed to produce the quads of a base on demand, packed with quad_id_bits
    // bits per model id below the base id
    void* cong;
    std::vector<unsigned long long> quad_off;
    int quad_id_bits;
    // candidates of the last stocs_make_transforms: device-resident (compacted, in pick order) as
    // T[n][16] | pose[n][16] | lcp[n] | base[n] inside d_cand; `cands` is the host mirror, filled on demand
    std::vector<stocs::Candidate> cands;
    char* d_cand;
    size_t cand_bytes;
    int n_cands, cand_cap;
    bool cands_stale;   // the device copy is newer than `cands`
    float best_lcp;
    int best_index;

    unsigned long long* d_best;   // 8-byte arg-max key
    bool best_is_zero;            // *d_best was zeroed on the stream by the last stocs_make_transforms and not used since

    // pinned host block for the small device-to-host read-backs of the entry points (totals, offsets, keys): a copy into
    // pageable memory goes through the runtime's own staging and is one more thing that can stall (ensure_pinned grows it)
    void* h_pin;
    size_t pin_bytes;
    stocs::CallTiming timing[4];   // last stocs_find_congruent_all / stocs_make_transforms / stocs_verify_all / stocs_run_trials

    stocs::StreamAudit audit;   // STOCS_DEBUG_STREAMS=1: happens-before check of the two-stream sections (stream_audit.h); off otherwise

    // scratch
    void* d_scratch;
    size_t scratch_bytes;
};

namespace stocs {
inline void clear_candidates(stocs_ctx* c) { c->cands.clear(); c->n_cands = 0; c->cands_stale = false; }
// the base set is (again) one trial's: whoever replaces or extends it outside stocs_run_trials calls this first
inline void clear_trial_batch(stocs_ctx* c) {
    if (c->base_seed.empty() && c->trial_first_base.empty()) return;
    c->base_seed.clear(); c->base_local.clear(); c->trial_first_base.clear(); c->trial_cand_off.clear();
    c->snrmw_trial0 = NULL; c->snrmw_stride = 0;
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);   // what is left of the batch's last piece means nothing to a single trial
}
inline float* cand_T(stocs_ctx* c) { return (float*)c->d_cand; }
inline float* cand_P(stocs_ctx* c) { return (float*)c->d_cand + (size_t)c->cand_cap * 16; }
inline float* cand_lcp(stocs_ctx* c) { return (float*)c->d_cand + (size_t)c->cand_cap * 32; }
inline int32_t* cand_base(stocs_ctx* c) { return (int32_t*)((float*)c->d_cand + (size_t)c->cand_cap * 33); }
int ensure_scratch(stocs_ctx* c, size_t bytes);
int ensure_pinned(stocs_ctx* c, size_t bytes);   // c->h_pin of at least `bytes` (nothing may still be copying into the old block)
enum { PIN_CONGRUENT = 0, PIN_TRANSFORMS = 256, PIN_VERIFY = 512, PIN_BEST = 768, PIN_VAR = 1024 };   // fixed slots, then the per-call variable part
// d_best8 != NULL: the kernel's epilogue also takes the arg-max of compute_best_transform over the batch into that word (zeroed in front)
int launch_lcp(stocs_ctx* c, const float* d_T16, int n, float* d_lcp, int32_t* d_hit, uint8_t* d_counted, unsigned long long* d_best8, uint32_t id_offset);
int sample_trials(stocs_ctx* c, int mode, int nT, const uint64_t* seeds, int nA, float dispersion, BaseOut* res_host, const float4** snrmw0, size_t* snrmw_stride);   // sample.hip
int build_ppf_index(stocs_ctx* c);
int build_grid_gpu(stocs_ctx* c, int div, int dense, int prune);
int prepare_cull_field(stocs_ctx* c);   // geometry + memory of SceneGrid::d_dist for the current grid and model (end of a grid build)
int fill_cull_field(stocs_ctx* c, hipStream_t st = NULL);      // the values, on st (NULL: c->stream); no synchronisation
extern "C" int stocs_internal_make_jobs(stocs_ctx* c, const int32_t* picks4_host, const int32_t* picks4_dev, int n, void* d_jobs_out, const unsigned int** d_unresolved_out);
extern "C" int stocs_internal_prepare_small(stocs_ctx* c, int max_per_base);   // small bases materialised while the host draws the picks
extern "C" void stocs_internal_free_congruent(stocs_ctx* c);
extern "C" void stocs_internal_invalidate_congruent(stocs_ctx* c);
extern "C" void stocs_internal_free_instance(stocs_ctx* c);
extern "C" void stocs_internal_free_trials(stocs_ctx* c);
// the congruent phase with a ceiling on its device memory: *too_big != 0 (and STOCS_OK) when the pair lists of the context's base set
// would need more than max_bytes (0: no ceiling) or exceed 2^32 entries -- a trial batch then splits the base set and tries again
extern "C" int stocs_internal_find_congruent(stocs_ctx* c, int64_t* total_quads, size_t max_bytes, int* too_big);
extern "C" void stocs_internal_invalidate_instance(stocs_ctx* c);
int plan_lookup(const PpfIndex& ix, const int* K, std::vector<std::pair<uint32_t, uint32_t> >* ranges);
void prefetch_lookup(const PpfIndex& ix, const int* K);
void compute_thresholds(const stocs_params& prm, Thresholds* t);
}  // namespace stocs

#endif
