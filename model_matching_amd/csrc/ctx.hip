// ctx.hip -- context construction for the MI355X StoCS engine.
// Replaces the ctor of stocs::stocs_estimator (reference include/stocs.hpp:18-61): clouds in,
// centroid_shift (reference src/stocs.cpp:943-964), spatial index over the scene (reference
// kdtree_initialize, stocs.cpp:966-980 -> here a brick grid, see SceneGrid), model normalisation
// into the unit cube (reference include/super4pcs/pairCreationFunctor.h:96-132).
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <limits>
#include <numeric>

#include "stocs_ctx.h"

namespace stocs {

static thread_local char g_err[512] = "";
unsigned long long g_dev_allocs = 0;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- do two streams run side by side? ----
// A context runs parts of stocs_find_congruent_all / stocs_make_transforms on its auxiliary stream next to the main one.  The runtime
// multiplexes the streams of a process onto a few hardware queues (4 per priority class by default, GPU_MAX_HW_QUEUES) and two streams on
// one queue run one after the other: with another context and torch's streams in the process, both streams of a context shared a queue
// and a Cm trial lost 13 % (bench.py's pipeline section: 5.9 M poses/s against 6.7 M in a trial-only process; GPU_MAX_HW_QUEUES=8 or 2, or
// closing the other context, brought it back -- round 4).  Another priority class for the auxiliary stream separates the two for certain
// but costs eight host threads with a context each a third of their throughput.  So the context asks the device: a one-thread kernel
// on the first stream waits (at most 100 us) for a flag that a kernel on the second sets; it sees the flag only if the second kernel ran
// while it was waiting.
__global__ void overlap_wait_kernel(unsigned int* flag, unsigned int* seen) {
    const unsigned long long t0 = wall_clock64();      // 100 MHz
    unsigned int v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < 10000ull) __builtin_amdgcn_s_sleep(8);
    *seen = v;
}
__global__ void overlap_set_kernel(unsigned int* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 1: the streams overlap, 0: they do not (or the probe could not run: the caller keeps what it has)
static int streams_run_side_by_side(hipStream_t a, hipStream_t b) {
    unsigned int* d = NULL;
    if (dev_malloc((void**)&d, 256) != hipSuccess) return 0;
    unsigned int h[2] = {0u, 0u};
    int ok = 0;
    if (hipMemsetAsync(d, 0, 8, a) == hipSuccess && hipStreamSynchronize(a) == hipSuccess) {
        hipLaunchKernelGGL(overlap_wait_kernel, dim3(1), dim3(1), 0, a, d, d + 1);
        hipLaunchKernelGGL(overlap_set_kernel, dim3(1), dim3(1), 0, b, d);
        if (hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess && hipMemcpy(h, d, 8, hipMemcpyDeviceToHost) == hipSuccess) ok = h[1] != 0u;
    }
    (void)hipGetLastError();
    (void)hipFree(d);
    return ok;
}

static inline int32_t float_ord(float f) {
    int32_t i;
    memcpy(&i, &f, 4);
    return i < 0 ? (int32_t)(0x80000000u - (uint32_t)i) : i;
}
static inline float ord_float(int32_t o) {
    int32_t i = o < 0 ? (int32_t)(0x80000000u - (uint32_t)o) : o;
    float f;
    memcpy(&f, &i, 4);
    return f;
}

// The acos-based predicates of the reference are monotone in the dot product, so each is an exact
// float threshold.  The thresholds are found by bisection against THIS host's libm (the same libm
// the reference binary would link), which reproduces the reference bit-for-bit without a device acos.
//   LCP   (stocs.cpp:1028-1032): float angle_n = std::acos(d)*180/M_PI; counted iff angle_n < A
//   base  (stocs.cpp:428-429,440): float a = acos(d)*180/M_PI; reject iff min(a, 180-a) < B
template <class Pred>
static float first_true_ascending(Pred pred) {  // pred false ... false true ... true on [-1, 1]
    int32_t lo = float_ord(-1.0f), hi = float_ord(1.0f);
    if (!pred(1.0f)) return std::numeric_limits<float>::infinity();
    if (pred(-1.0f)) return -1.0f;
    while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        if (pred(ord_float(mid))) hi = mid; else lo = mid;
    }
    return ord_float(hi);
}
template <class Pred>
static float last_true_descending(Pred pred) {  // pred true ... true false ... false on [-1, 1]
    int32_t lo = float_ord(-1.0f), hi = float_ord(1.0f);
    if (!pred(-1.0f)) return -std::numeric_limits<float>::infinity();
    if (pred(1.0f)) return 1.0f;
    while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        if (pred(ord_float(mid))) lo = mid; else hi = mid;
    }
    return ord_float(lo);
}

void compute_thresholds(const stocs_params& prm, Thresholds* t) {
    const float A = prm.lcp_normal_angle;
    t->lcp_dot_lo = first_true_ascending([A](float d) {
        float angle_n = (float)((double)(acosf(d) * 180) / M_PI);
        return angle_n < A;
    });
    const float B = prm.internal_angle_threshold;
    t->ang_dot_hi = first_true_ascending([B](float d) {
        float a = (float)(acos((double)d) * 180 / M_PI);
        return a < B;
    });
    t->ang_dot_lo = last_true_descending([B](float d) {
        float a = (float)(acos((double)d) * 180 / M_PI);
        float o = 180 - a;
        return o < B;
    });
}

int ensure_scratch(stocs_ctx* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return STOCS_OK;
    if (c->d_scratch) STOCS_HIP_CHECK(hipFree(c->d_scratch));
    c->d_scratch = NULL;
    c->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + (1 << 20);
    STOCS_HIP_CHECK(dev_malloc(&c->d_scratch, want));
    c->scratch_bytes = want;
    return STOCS_OK;
}

int ensure_pinned(stocs_ctx* c, size_t bytes) {
    if (c->pin_bytes >= bytes) return STOCS_OK;
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = NULL; c->pin_bytes = 0; }
    const size_t cap = std::max<size_t>(2 * bytes, (size_t)64 << 10);
    STOCS_HIP_CHECK(pinned_malloc(&c->h_pin, cap));
    c->pin_bytes = cap;
    return STOCS_OK;
}

static inline uint32_t part1by2(uint32_t x) {
    x &= 0x3ff;
    x = (x | (x << 16)) & 0x30000ff;
    x = (x | (x << 8)) & 0x300f00f;
    x = (x | (x << 4)) & 0x30c30c3;
    x = (x | (x << 2)) & 0x9249249;
    return x;
}

template <class T>
static int upload(T** dptr, const T* h, size_t n) {
    *dptr = NULL;
    if (n == 0) n = 1;
    STOCS_HIP_CHECK(dev_malloc((void**)dptr, n * sizeof(T)));
    if (h) STOCS_HIP_CHECK(hipMemcpy(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice));
    return STOCS_OK;
}

static void free_grid(stocs_ctx* c) {   // the grid lives in c->grid_mem, which the next build resets
    c->grid.d_top = NULL; c->grid.d_cells = NULL; c->grid.d_list = NULL; c->grid.d_chunk_r = NULL; c->grid.d_flat = NULL;
    c->grid.d_dist = NULL; c->grid.dist_ready = false;
}

// One build at cell edge eps / div.  At eps the index-ordered lists with the flat cell table and the sub-cell masks (the sparse
// layout); finer grids have neither, and the centre-sorted lists with early exit (the dense layout) win there whatever the list
// length (65 000-point scene at eps/2: 1.53 ms against 1.99 for index-ordered lists of 13 entries; tools/layout_sweep.py).
// last_resort: eps is final although its lists are long (the finer grid does not fit): centre-sorted there too.
static int build_grid_once(stocs_ctx* c, int div, bool last_resort = false) {
    // STOCS_GRID_DENSE (measurement switch, tools/layout_sweep.py): 0 = index-ordered lists whatever their length, 1 = centre-sorted always
    const char* force = getenv("STOCS_GRID_DENSE");
    // round 5: dominance-pruned lists (grid.hip) -- index-ordered at every cell edge, no early exit needed; "grid_prune" 0 / STOCS_GRID_PRUNE=0
    // keeps the layouts of rounds 2-4 (A/B and cross-check: same scores bit for bit)
    const char* pe = getenv("STOCS_GRID_PRUNE");
    const int prune = pe ? atoi(pe) : c->grid_prune;
    if (prune && !force) return build_grid_gpu(c, div, 0, 1);
    if (force) return build_grid_gpu(c, div, atoi(force) == 1 ? 1 : 0, 0);
    if (div > 1) return build_grid_gpu(c, div, 1, 0);
    int rc = build_grid_gpu(c, div, 0, 0);
    if (rc || !last_resort || c->grid.avg_list_len <= 16.0) return rc;
    free_grid(c);
    return build_grid_gpu(c, div, 1, 0);
}

static int build_grid_levels(stocs_ctx* c) {
    int div = c->grid_div;
    const char* e = getenv("STOCS_GRID_DIV");
    if (e) div = atoi(e);
    const bool fixed = e || c->grid_div != 1;
    int rc = build_grid_once(c, div, fixed);
    auto report = [&]() {
        if (getenv("STOCS_DEBUG_TIMING"))
            fprintf(stderr, "[stocs grid] cell edge eps/%d, %d bricks, %lld list entries (%.1f per non-empty cell; %.1f within r of the cell%s)\n", (int)lround((double)c->prm.distance_threshold / c->grid.h),
                    c->grid.n_bricks, (long long)c->grid.n_entries, c->grid.avg_list_len, c->grid.avg_dilated_len, c->grid.pruned ? ", dominance-pruned" : "");
    };
    if (rc || fixed) { if (!rc) report(); return rc; }
    for (int next = 2; next <= 4; next *= 2) {
        // Stop when the lists are short, or when the finer grid would not fit comfortably (entries x ~8, 16 B each).  Thresholds
        // (second half of round 3, tools/layout_sweep.py, 16 384-32 768 candidates; entries per non-empty list at the coarser edge):
        //   eps -> eps/2 from 12.5 on: 10.2: 1.22 (eps) vs 1.24 ms (eps/2); 12.5: 1.68 vs 1.64; 15.8: 2.40 vs 2.14;
        //   eps/2 -> eps/4 from 18 on: 13.1: 1.51 (eps/2) vs 1.68 (eps/4); 16.2: 2.08 vs 2.09; 20.9: 2.73 vs 2.60; 140 000 points: 4.39 vs 3.76
        // (rounds 1-3a: 16 and 28, with the sparse layout kept at eps/2 for lists up to 16 entries)
        // Round 5, dominance-pruned lists (tools/prune_layout_sweep.sh, 16 384 candidates, ms at eps / eps/2 / eps/4): 20 000 points 0.32 / 0.41 /
        // 0.55; 35 000: 0.48 / 0.57 / 0.85; 50 000: 0.83 / 0.79 / 1.14; 65 000: 1.17 / 0.96 / 1.41; 100 000: 2.30 / 1.46 / 2.16; 140 000: 4.24 /
        // 2.16 / 2.95; C5 (200 000): 8.49 / 3.40 / 4.09 -- eps/2 from the same density on as before (the count of points within r of a cell at
        // eps: 12.5), eps/4 no longer at any of these densities: a pruned list at eps/2 holds what can win somewhere in a 2.5 mm cell, and
        // the eight times as many cell words of eps/4 cost more than its shorter lists save.  eps/4 stays for pruned lists beyond 16
        // entries at eps/2 (scenes several times denser than C5; not measured).
        const double n_inc_now = c->grid.avg_dilated_len / std::max(c->grid.avg_list_len, 1.0) * (double)c->grid.n_entries;   // ~ the (cell, point) incidences of this build
        const bool finer_pays = c->grid.pruned ? (next == 2 ? c->grid.avg_dilated_len > 12.5 : c->grid.avg_list_len > 16.0)
                                               : c->grid.avg_dilated_len > (next == 2 ? 12.5 : 18.0);
        if (!finer_pays || (int64_t)n_inc_now * 8 >= ((int64_t)1 << 29)) break;
        const int prev = next / 2;
        free_grid(c);
        rc = build_grid_once(c, next);
        if (rc == STOCS_ERR_INVALID) {   // the finer grid does not fit the 32-bit list offsets: stay with the coarser one
            free_grid(c);
            rc = build_grid_once(c, prev, true);
            break;
        }
        if (rc) return rc;
    }
    if (!rc) report();
    return rc;
}

static int build_grid(stocs_ctx* c) {
    int rc = build_grid_levels(c);
    if (!rc) rc = prepare_cull_field(c);   // geometry + memory only; filled when scoring calls make it pay (lcp.hip)
    return rc;
}

// Everything that depends on the scene cloud: host copies (kdtree_initialize / centroid_shift of the scene,
// stocs.cpp:943-980), device clouds, the brick grid; per-trial state is reset.  Used by stocs_ctx_create and
// stocs_ctx_set_scene (a new camera frame against the same model keeps the model clouds and the PPF index).
static int load_scene_impl(stocs_ctx* c, const float* sp, const float* sn, const float* sprob, const int32_t* spix, int nS) {
    const bool dbg_t = getenv("STOCS_DEBUG_TIMING") != NULL;
    struct timespec ts_a; clock_gettime(CLOCK_MONOTONIC, &ts_a);
    auto lap = [&](const char* what) {
        if (!dbg_t) return;
        struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
        fprintf(stderr, "[stocs scene] %-22s %8.3f ms\n", what, (t.tv_sec - ts_a.tv_sec) * 1e3 + (t.tv_nsec - ts_a.tv_nsec) * 1e-6);
        ts_a = t;
    };
    // a frame stream on a fixed camera: when the frame before this one was scored often enough for its distance field to pay, this
    // one will be too -- its field is filled at once, next to whatever the caller does first (sampling), instead of after a warm-up
    c->prev_scene_warm = c->lcp_cull == 1 && c->grid.d_dist != NULL && c->scene_work >= c->lcp_cull_after;
    if (c->cull_pending && c->aux_stream) { STOCS_HIP_CHECK(hipStreamSynchronize(c->aux_stream)); c->cull_pending = false; }   // the old field's memory is about to be recycled
    c->nS = nS;
    c->h_spos.resize(nS); c->h_snrm.resize(nS); c->h_sprob.assign(sprob, sprob + nS); c->h_sprob0 = c->h_sprob; c->h_spix.assign((size_t)2 * nS, 0);
    for (int i = 0; i < nS; ++i) {
        c->h_spos[i] = mk3(sp[3 * i], sp[3 * i + 1], sp[3 * i + 2]);
        c->h_snrm[i] = normalized3(mk3(sn[3 * i], sn[3 * i + 1], sn[3 * i + 2]));  // set_normal, point3d.hpp:43-45
        if (spix) { c->h_spix[2 * i] = spix[2 * i]; c->h_spix[2 * i + 1] = spix[2 * i + 1]; }
    }
    // centroid_shift -- stocs.cpp:943-964 (sequential float sums, then divide, then subtract)
    V3 cs = mk3(0, 0, 0);
    for (int i = 0; i < nS; ++i) cs = cs + c->h_spos[i];
    cs = cs / (float)nS;
    for (int i = 0; i < nS; ++i) c->h_spos[i] = c->h_spos[i] - cs;
    c->centroid_scene = cs;
    free_grid(c);
    lap("host copies + centroid");
    int rc = STOCS_OK;
    {
        // one slab, grown only when a frame has more points than any before it: a camera stream does no hipMalloc / hipFree
        // per frame (hipFree alone synchronises the whole device), and one copy moves the three arrays
        const size_t n = (size_t)std::max(nS, 1);
        if (n > c->scene_cap) {
            STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->d_scene_mem) { (void)hipFree(c->d_scene_mem); c->d_scene_mem = NULL; c->scene_cap = 0; }
            const size_t cap = n + n / 4 + 256;
            STOCS_HIP_CHECK(dev_malloc((void**)&c->d_scene_mem, cap * 40));
            c->scene_cap = cap;
        }
        c->d_spos = (float4*)c->d_scene_mem;
        c->d_snrmw = (float4*)(c->d_scene_mem + c->scene_cap * 16);
        c->d_spix = (int2*)(c->d_scene_mem + c->scene_cap * 32);
        // the three arrays are laid out in the context's PINNED block and go up from there (round 5b: two std::vectors of half a megabyte, three
        // copies out of pageable memory and a synchronisation were ~80 us of a frame's stocs_ctx_set_scene); the grid build's first
        // synchronisation is behind them before anything else touches the block
        { const int rcp = ensure_pinned(c, (size_t)PIN_VAR + n * 40 + 256); if (rcp) return rcp; }
        float4* ab = (float4*)((char*)c->h_pin + PIN_VAR);
        int2* px = (int2*)((char*)c->h_pin + PIN_VAR + n * 32);
        for (int i = 0; i < nS; ++i) {
            ab[i] = make_float4(c->h_spos[i].x, c->h_spos[i].y, c->h_spos[i].z, c->h_sprob[i]);
            ab[n + i] = make_float4(c->h_snrm[i].x, c->h_snrm[i].y, c->h_snrm[i].z, c->h_sprob[i]);
            px[i] = make_int2(c->h_spix[2 * i], c->h_spix[2 * i + 1]);
        }
        // stream-ordered behind whatever still reads the old frame
        c->prior_epoch++;
        STOCS_HIP_CHECK(hipMemcpyAsync(c->d_spos, ab, n * 16, hipMemcpyHostToDevice, c->stream));
        STOCS_HIP_CHECK(hipMemcpyAsync(c->d_snrmw, ab + n, n * 16, hipMemcpyHostToDevice, c->stream));
        STOCS_HIP_CHECK(hipMemcpyAsync(c->d_spix, px, n * 8, hipMemcpyHostToDevice, c->stream));
        STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    lap("upload");
    if (!rc) rc = build_grid(c);
    lap("grid");
    if (!rc && c->prev_scene_warm && c->lcp_cull == 1 && c->grid.d_dist && c->d_mpatch && c->aux_stream) {
        // (the grid build has synchronised c->stream: the scene arrays the fill reads are in place)
        rc = fill_cull_field(c, c->aux_stream);
        if (!rc) { STOCS_HIP_CHECK(hipEventRecord(c->ev_cull, c->aux_stream)); c->cull_pending = true; }
    }
    // per-trial state belongs to the old scene
    clear_trial_batch(c);
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1;
    c->scene_scored = 0; c->scene_work = 0.0;
    stocs_internal_invalidate_congruent(c);
    const size_t npx = (size_t)c->prm.image_width * c->prm.image_height;
    c->has_edge = false;
    c->edge_map.assign(npx, 0);
    stocs_internal_invalidate_instance(c);
    c->last_segment.clear();
    return rc;
}

// A failed load (allocation, copy or grid build) must not leave a context that still points at the previous frame's --
// possibly freed -- device clouds: it becomes scene-less (nS = 0: sampling reports no bases, scoring refuses) until the
// next successful stocs_ctx_set_scene.
static int load_scene(stocs_ctx* c, const float* sp, const float* sn, const float* sprob, const int32_t* spix, int nS) {
    if (spix) {   // instance-mode sampling indexes the 2-D maps with these (sample.hip); refuse what would land outside them
        const int W = c->prm.image_width, H = c->prm.image_height;   // (a refused frame changes nothing: the old scene stays)
        for (int i = 0; i < nS; ++i)
            if (spix[2 * i] < 0 || spix[2 * i] >= H || spix[2 * i + 1] < 0 || spix[2 * i + 1] >= W) {
                set_error("scene point %d has pixel (row %d, col %d) outside the %dx%d image of stocs_params", i, spix[2 * i], spix[2 * i + 1], W, H);
                return STOCS_ERR_INVALID;
            }
    }
    const int rc = load_scene_impl(c, sp, sn, sprob, spix, nS);
    if (rc != STOCS_OK) {
        c->nS = 0;
        c->h_spos.clear(); c->h_snrm.clear(); c->h_sprob.clear(); c->h_sprob0.clear(); c->h_spix.clear();
        c->d_spos = NULL; c->d_snrmw = NULL; c->d_spix = NULL;
        free_grid(c);
        c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
        c->best_lcp = 0; c->best_index = -1;
        stocs_internal_invalidate_congruent(c);
        stocs_internal_invalidate_instance(c);
        c->last_segment.clear();
    }
    return rc;
}

// Order of the centred model for the scan kernels and the bounding sphere of every 64-point step (host only: no device involved;
// stocs_model_patch_order exposes it to the CPU tests).  mpos: centred positions, munit: unit-cube positions (Morton order only).
static void model_patch_order(const std::vector<V3>& mpos, const std::vector<V3>& munit, std::vector<int32_t>& perm, std::vector<float4>& patch, float* r_ref) {
    const int nM = (int)mpos.size();
    // Order of the centred model for the LCP kernel: 64 consecutive points = one step of a wavefront = one compact surface patch
    // (spatially coherent look-ups, and a small bounding sphere for the patch test).  Median splits along the longest axis with
    // the left part a multiple of 64 points: every leaf is one step, neighbouring leaves are neighbouring patches.
    // (Rounds 1-2 used the Morton order of the unit-cube coordinates: patches of 25 mm radius in the median on the 5 000-point
    // model against 19 mm here; STOCS_MODEL_ORDER=morton keeps it selectable for the A/B.)
    perm.resize(nM);
    std::iota(perm.begin(), perm.end(), 0);
    if (getenv("STOCS_MODEL_ORDER") && !strcmp(getenv("STOCS_MODEL_ORDER"), "morton")) {
        std::vector<uint32_t> code(nM);
        for (int i = 0; i < nM; ++i) {
            const V3 u = munit[i];
            auto q10 = [](float v) { int k = (int)(v * 1024.0f); return (uint32_t)(k < 0 ? 0 : (k > 1023 ? 1023 : k)); };
            code[i] = (part1by2(q10(u.z)) << 2) | (part1by2(q10(u.y)) << 1) | part1by2(q10(u.x));
        }
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return code[a] < code[b]; });
    } else {
        struct Range { int lo, hi; };
        std::vector<Range> todo(1, Range{0, nM});
        while (!todo.empty()) {
            const Range r = todo.back(); todo.pop_back();
            const int n = r.hi - r.lo;
            if (n <= 64) continue;
            float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
            for (int k = r.lo; k < r.hi; ++k) {
                const V3 q = mpos[perm[k]];
                const float v[3] = {q.x, q.y, q.z};
                for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], v[a]); mx[a] = std::max(mx[a], v[a]); }
            }
            int ax = 0;
            for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
            const int leaves = (n + 63) / 64, nl = (leaves / 2) * 64;   // >= 64, < n
            auto coord = [&](int id) { const V3 q = mpos[id]; return ax == 0 ? q.x : (ax == 1 ? q.y : q.z); };
            std::stable_sort(perm.begin() + r.lo, perm.begin() + r.hi, [&](int a, int b) { return coord(a) < coord(b); });
            todo.push_back(Range{r.lo + nl, r.hi});
            todo.push_back(Range{r.lo, r.lo + nl});
        }
    }
    // bounding sphere per 64-point step (double arithmetic; centre = a few steps of Ritter's iteration towards the farthest point,
    // radius = the exact maximum distance from that centre, rounded up)
    const int n_patch = (nM + 63) / 64;
    patch.assign((size_t)std::max(n_patch, 1), make_float4(0.f, 0.f, 0.f, 0.f));
    {
        std::vector<float> radii;
        for (int s = 0; s < n_patch; ++s) {
            const int lo = 64 * s, hi = std::min(nM, lo + 64);
            double ctr[3] = {0, 0, 0};
            for (int k = lo; k < hi; ++k) { const V3 q = mpos[perm[k]]; ctr[0] += q.x; ctr[1] += q.y; ctr[2] += q.z; }
            for (int a = 0; a < 3; ++a) ctr[a] /= (double)(hi - lo);
            auto farthest = [&](double* d_out) {
                int best = lo; double bd = -1;
                for (int k = lo; k < hi; ++k) {
                    const V3 q = mpos[perm[k]];
                    const double dx = q.x - ctr[0], dy = q.y - ctr[1], dz = q.z - ctr[2], d = dx * dx + dy * dy + dz * dz;
                    if (d > bd) { bd = d; best = k; }
                }
                *d_out = sqrt(bd);
                return best;
            };
            double rad = 0;
            for (int it = 0; it < 64; ++it) {
                const int f = farthest(&rad);
                const V3 q = mpos[perm[f]];
                const double step = 0.5 / (double)(it + 2);
                ctr[0] += (q.x - ctr[0]) * step; ctr[1] += (q.y - ctr[1]) * step; ctr[2] += (q.z - ctr[2]) * step;
            }
            const float cf[3] = {(float)ctr[0], (float)ctr[1], (float)ctr[2]};
            ctr[0] = cf[0]; ctr[1] = cf[1]; ctr[2] = cf[2];   // the radius belongs to the centre as stored
            (void)farthest(&rad);
            const float rf = (float)(rad * (1.0 + 1e-6) + 1e-7);
            patch[s] = make_float4(cf[0], cf[1], cf[2], rf);
            radii.push_back(rf);
        }
        std::sort(radii.begin(), radii.end());
        *r_ref = radii.empty() ? 0.0f : radii[(size_t)((radii.size() - 1) * 0.8)];
    }

}

}  // namespace stocs

using namespace stocs;

extern "C" {

const char* stocs_last_error(void) { return g_err; }
const char* stocs_version(void) { return "stocs_hip 0.1 (gfx950)"; }

void stocs_default_params(stocs_params* p) {
    p->distance_threshold = 0.005f;
    p->ppf_tr_discretization = 5;
    p->ppf_rot_discretization = 5;
    p->plane_threshold = 0.015f;
    p->min_distance_base = 0.01f;
    p->internal_angle_threshold = 30.0f;
    p->lcp_normal_angle = 30.0f;
    p->image_width = 640;
    p->image_height = 480;
    p->number_of_bases = 100;
    p->maximum_congruent_sets = 200;
}

int stocs_ctx_create(const stocs_params* prm, const float* sp, const float* sn, const float* sprob,
                     const int32_t* spix, int nS, const float* mp, const float* mn, int nM, int build_index,
                     int device, stocs_ctx** out) {
    if (!prm || !out || nS < 0 || nM < 0 || (nS && (!sp || !sn || !sprob)) || (nM && (!mp || !mn))) {
        set_error("stocs_ctx_create: invalid argument");
        return STOCS_ERR_INVALID;
    }
    if (nS == 0 || nM == 0) {  // the reference indexes an empty vector here (stocs.cpp:386); refuse instead
        set_error("stocs_ctx_create: empty scene or model cloud");
        return STOCS_ERR_INVALID;
    }
    if (nM > 65535) { set_error("model has %d points; at most 65535 supported (16-bit ids in packed pairs)", nM); return STOCS_ERR_INVALID; }
    if (prm->ppf_rot_discretization <= 0 || prm->ppf_tr_discretization <= 0 || 180 % prm->ppf_rot_discretization) {
        set_error("PPF discretisation must be positive and divide 180");
        return STOCS_ERR_INVALID;
    }
    *out = NULL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: this library has no CPU fallback");
        return STOCS_ERR_NO_DEVICE;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return STOCS_ERR_NO_DEVICE; }
    DeviceGuard dev_guard(device);   // the caller's current device is restored on return
    { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != device) { set_error("hipSetDevice(%d) failed", device); return STOCS_ERR_NO_DEVICE; } }

    stocs_ctx* c = new stocs_ctx();
    c->prm = *prm;
    c->device = device;
    c->nS = nS; c->nM = nM;
    c->d_scratch = NULL; c->scratch_bytes = 0;
    c->h_pin = NULL; c->pin_bytes = 0;
    for (int k = 0; k < 4; ++k) c->timing[k].n = 0;
    c->index.built = false;
    c->index.d_bucket_start = NULL; c->index.d_pairs = NULL; c->index.d_exists = NULL;
    c->cong = NULL; c->quad_id_bits = 16;
    c->inst = NULL;
    c->trials = NULL; c->snrmw_trial0 = NULL; c->snrmw_stride = 0; c->snrmw_override = NULL; c->lcp_cand_trial = NULL;
    c->d_cand = NULL; c->cand_bytes = 0; c->n_cands = 0; c->cand_cap = 0; c->cands_stale = false;
    c->d_best = NULL; c->best_is_zero = false;
    c->best_lcp = 0; c->best_index = -1;
    c->has_edge = false;
    c->grid_div = 1;
    c->grid_prune = getenv("STOCS_GRID_PRUNE") ? atoi(getenv("STOCS_GRID_PRUNE")) : 1;
    c->lcp_variant = -1;
    c->device_clock = (getenv("STOCS_DEVICE_CLOCK") && atoi(getenv("STOCS_DEVICE_CLOCK")) != 0) ? 1 : 0;
    c->lcp_split = 1;
    c->lcp_flat = getenv("STOCS_LCP_FLAT") ? atoi(getenv("STOCS_LCP_FLAT")) : 1;
    c->lcp_order = getenv("STOCS_LCP_ORDER") ? atoi(getenv("STOCS_LCP_ORDER")) : 1;
    c->d_order = NULL; c->order_bytes = 0;
    c->d_cdf = NULL; c->cdf_bytes = 0; c->cdf_n = 0; c->prior_epoch = 1; c->cdf_epoch = 0;
    memset(&c->grid, 0, sizeof(c->grid));
    c->d_spos = c->d_snrmw = c->d_mpos = c->d_mnrm = c->d_munit = c->d_mpos_raw = c->d_mpos_s = c->d_mnrm_s = NULL;
    c->d_spix = NULL; c->d_mperm = NULL; c->d_mpatch = NULL; c->d_scene_mem = NULL; c->scene_cap = 0;
    c->patch_r_ref = 0.0f; c->scene_scored = 0; c->scene_work = 0.0;
    c->lcp_group = getenv("STOCS_LCP_GROUP") ? atoi(getenv("STOCS_LCP_GROUP")) : 4;
    c->lcp_cull = getenv("STOCS_LCP_CULL") ? atoi(getenv("STOCS_LCP_CULL")) : 1;
    c->lcp_cull_after = 1.0e9; c->prev_scene_warm = false; c->cull_pending = false; c->ev_cull = NULL;
    c->stream = NULL; c->own_stream = NULL; c->aux_stream = NULL;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_cull, hipEventDisableTiming) != hipSuccess) {
        set_error("stream/event creation failed");
        delete c;
        return STOCS_ERR_NO_DEVICE;
    }
    for (int k = 0; k < 10; ++k)
        if (hipEventCreate(&c->ev_t[k]) != hipSuccess) { set_error("event creation failed"); delete c; return STOCS_ERR_NO_DEVICE; }
    if (!getenv("STOCS_NO_STREAM_PROBE")) {   // an auxiliary stream that runs NEXT to the main one (see streams_run_side_by_side): up to four candidates
        hipStream_t spare[4]; int n_spare = 0;
        while (!streams_run_side_by_side(c->own_stream, c->aux_stream) && n_spare < 4) {
            hipStream_t s2 = NULL;
            if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            spare[n_spare++] = c->aux_stream;      // (kept alive until the choice is made: the runtime hands the next stream another queue)
            c->aux_stream = s2;
        }
        for (int k = 0; k < n_spare; ++k) (void)hipStreamDestroy(spare[k]);
        if (getenv("STOCS_DEBUG_TIMING")) fprintf(stderr, "[stocs ctx] auxiliary stream: candidate %d runs next to the main stream\n", n_spare + 1);
    }
    c->stream = c->own_stream;
    compute_thresholds(c->prm, &c->thr);

    c->h_mpos.resize(nM); c->h_mnrm.resize(nM); c->h_mpos_raw.resize(nM);
    for (int i = 0; i < nM; ++i) {
        c->h_mpos_raw[i] = c->h_mpos[i] = mk3(mp[3 * i], mp[3 * i + 1], mp[3 * i + 2]);
        c->h_mnrm[i] = normalized3(mk3(mn[3 * i], mn[3 * i + 1], mn[3 * i + 2]));
    }
    // centroid_shift -- stocs.cpp:943-964 (sequential float sums, then divide, then subtract)
    V3 cm = mk3(0, 0, 0);
    for (int i = 0; i < nM; ++i) cm = cm + c->h_mpos[i];
    cm = cm / (float)nM;
    for (int i = 0; i < nM; ++i) c->h_mpos[i] = c->h_mpos[i] - cm;
    c->centroid_model = cm;

    // PairCreationFunctor::synch3DContent -- pairCreationFunctor.h:96-132 (once, not per base)
    {
        const float big = std::numeric_limits<float>::max() / 2;
        V3 bmn = mk3(big, big, big), bmx = mk3(-big, -big, -big);
        for (int i = 0; i < nM; ++i) {
            const V3 q = c->h_mpos[i];
            if (q.x < bmn.x) bmn.x = q.x; if (q.y < bmn.y) bmn.y = q.y; if (q.z < bmn.z) bmn.z = q.z;
            if (q.x > bmx.x) bmx.x = q.x; if (q.y > bmx.y) bmx.y = q.y; if (q.z > bmx.z) bmx.z = q.z;
        }
        c->gcenter = bmn + ((bmx - bmn) / 2.0f);
        const V3 ext = bmx - bmn;
        const double r = std::max((double)ext.z + 0.001, std::max((double)ext.y + 0.001, (double)ext.x + 0.001));
        c->ratio = (float)r;
        c->h_munit.resize(nM);
        const V3 half = mk3(0.5f, 0.5f, 0.5f);
        for (int i = 0; i < nM; ++i) c->h_munit[i] = (c->h_mpos[i] - c->gcenter) / c->ratio + half;
    }

    std::vector<float4> patch;
    model_patch_order(c->h_mpos, c->h_munit, c->h_mperm, patch, &c->patch_r_ref);
    const int n_patch = (nM + 63) / 64;

    int rc = STOCS_OK;
    {
        const int n = std::max(nM, 1);
        std::vector<float4> a(n), b(n), u(n), raw(n), as(n), bs(n);
        for (int i = 0; i < nM; ++i) {
            a[i] = make_float4(c->h_mpos[i].x, c->h_mpos[i].y, c->h_mpos[i].z, 0.f);
            b[i] = make_float4(c->h_mnrm[i].x, c->h_mnrm[i].y, c->h_mnrm[i].z, 0.f);
            u[i] = make_float4(c->h_munit[i].x, c->h_munit[i].y, c->h_munit[i].z, 0.f);
            raw[i] = make_float4(c->h_mpos_raw[i].x, c->h_mpos_raw[i].y, c->h_mpos_raw[i].z, 0.f);
        }
        for (int i = 0; i < nM; ++i) { as[i] = a[c->h_mperm[i]]; bs[i] = b[c->h_mperm[i]]; }
        // the sorted positions are padded to whole 64-point steps with NaN: the scan kernels load and transform a step without
        // bounds checks, and a NaN query matches nothing (every comparison of its distances fails)
        as.resize((size_t)n_patch * 64 + 64, make_float4(NAN, NAN, NAN, 0.f));
        if (!rc) rc = upload(&c->d_mpos, a.data(), a.size());
        if (!rc) rc = upload(&c->d_mnrm, b.data(), b.size());
        if (!rc) rc = upload(&c->d_munit, u.data(), u.size());
        if (!rc) rc = upload(&c->d_mpos_raw, raw.data(), raw.size());
        if (!rc) rc = upload(&c->d_mpos_s, as.data(), as.size());
        if (!rc) rc = upload(&c->d_mnrm_s, bs.data(), bs.size());
        if (!rc) rc = upload(&c->d_mperm, c->h_mperm.data(), c->h_mperm.size());
        if (!rc) rc = upload(&c->d_mpatch, patch.data(), patch.size());
    }
    if (!rc) rc = load_scene(c, sp, sn, sprob, spix, nS);
    if (!rc && build_index) rc = build_ppf_index(c);
    if (rc) { stocs_ctx_destroy(c); return rc; }
    *out = c;
    return STOCS_OK;
}

int stocs_model_patch_order(const float* model_pos3, int nM, int32_t* perm, float* patches4) {
    if (!model_pos3 || nM <= 0 || !perm || !patches4) return STOCS_ERR_INVALID;
    // centroid_shift and the unit cube exactly as stocs_ctx_create does them (stocs.cpp:943-964, pairCreationFunctor.h:96-132)
    std::vector<V3> mpos(nM), munit(nM);
    V3 cm = mk3(0, 0, 0);
    for (int i = 0; i < nM; ++i) { mpos[i] = mk3(model_pos3[3 * i], model_pos3[3 * i + 1], model_pos3[3 * i + 2]); cm = cm + mpos[i]; }
    cm = cm / (float)nM;
    const float big = std::numeric_limits<float>::max() / 2;
    V3 bmn = mk3(big, big, big), bmx = mk3(-big, -big, -big);
    for (int i = 0; i < nM; ++i) {
        mpos[i] = mpos[i] - cm;
        const V3 q = mpos[i];
        if (q.x < bmn.x) bmn.x = q.x; if (q.y < bmn.y) bmn.y = q.y; if (q.z < bmn.z) bmn.z = q.z;
        if (q.x > bmx.x) bmx.x = q.x; if (q.y > bmx.y) bmx.y = q.y; if (q.z > bmx.z) bmx.z = q.z;
    }
    const V3 gcenter = bmn + ((bmx - bmn) / 2.0f), ext = bmx - bmn;
    const float ratio = (float)std::max((double)ext.z + 0.001, std::max((double)ext.y + 0.001, (double)ext.x + 0.001));
    for (int i = 0; i < nM; ++i) munit[i] = (mpos[i] - gcenter) / ratio + mk3(0.5f, 0.5f, 0.5f);
    std::vector<int32_t> pm; std::vector<float4> patch; float r_ref = 0.0f;
    model_patch_order(mpos, munit, pm, patch, &r_ref);
    for (int i = 0; i < nM; ++i) perm[i] = pm[i];
    for (int s = 0; s < (nM + 63) / 64; ++s) { patches4[4 * s] = patch[s].x; patches4[4 * s + 1] = patch[s].y; patches4[4 * s + 2] = patch[s].z; patches4[4 * s + 3] = patch[s].w; }
    return STOCS_OK;
}

int stocs_ctx_destroy(stocs_ctx* c) {
    if (!c) return STOCS_OK;
    DeviceGuard dev_guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->d_scene_mem, c->d_mpos, c->d_mnrm, c->d_munit, c->d_mpos_raw, c->d_mpos_s,
                    c->d_mnrm_s, c->d_mperm, c->d_mpatch, c->index.d_bucket_start,
                    c->index.d_pairs, c->index.d_exists, c->d_scratch, c->d_best, c->d_cand, c->d_order, c->d_cdf};
    stocs_internal_free_congruent(c);
    stocs_internal_free_instance(c);
    stocs_internal_free_trials(c);
    c->grid_mem.destroy(); c->grid_ws.destroy();
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    for (void* p : ptrs) if (p) (void)hipFree(p);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    if (c->ev_cull) (void)hipEventDestroy(c->ev_cull);
    for (int k = 0; k < 10; ++k) (void)hipEventDestroy(c->ev_t[k]);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return STOCS_OK;
}

int stocs_ctx_set_scene(stocs_ctx* c, const float* sp, const float* sn, const float* sprob, const int32_t* spix, int nS) {
    if (!c || nS <= 0 || !sp || !sn || !sprob) { set_error("stocs_ctx_set_scene: invalid argument"); return STOCS_ERR_INVALID; }
    DeviceGuard dev_guard(c->device);
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));   // nothing of the old scene may still be in flight
    return load_scene(c, sp, sn, sprob, spix, nS);
}

int stocs_get_centroids(const stocs_ctx* c, float* s, float* m) {
    if (!c) return STOCS_ERR_INVALID;
    if (s) { s[0] = c->centroid_scene.x; s[1] = c->centroid_scene.y; s[2] = c->centroid_scene.z; }
    if (m) { m[0] = c->centroid_model.x; m[1] = c->centroid_model.y; m[2] = c->centroid_model.z; }
    return STOCS_OK;
}
int stocs_get_scene(const stocs_ctx* c, float* pos3, float* nrm3, float* prob, int32_t* pix2) {
    if (!c) return STOCS_ERR_INVALID;
    for (int i = 0; i < c->nS; ++i) {
        if (pos3) { pos3[3 * i] = c->h_spos[i].x; pos3[3 * i + 1] = c->h_spos[i].y; pos3[3 * i + 2] = c->h_spos[i].z; }
        if (nrm3) { nrm3[3 * i] = c->h_snrm[i].x; nrm3[3 * i + 1] = c->h_snrm[i].y; nrm3[3 * i + 2] = c->h_snrm[i].z; }
        if (prob) prob[i] = c->h_sprob[i];
        if (pix2) { pix2[2 * i] = c->h_spix[2 * i]; pix2[2 * i + 1] = c->h_spix[2 * i + 1]; }
    }
    return STOCS_OK;
}
int stocs_get_sizes(const stocs_ctx* c, int* nS, int* nM) {
    if (!c) return STOCS_ERR_INVALID;
    if (nS) *nS = c->nS;
    if (nM) *nM = c->nM;
    return STOCS_OK;
}
int stocs_set_edge_map(stocs_ctx* c, const uint8_t* edge) {
    if (!c || !edge) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    c->edge_map.assign(edge, edge + (size_t)c->prm.image_width * c->prm.image_height);
    c->has_edge = true;
    stocs_internal_invalidate_instance(c);   // runs and per-point tables follow the edge map
    return STOCS_OK;
}

int stocs_sync(stocs_ctx* c) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}
void* stocs_stream(stocs_ctx* c) { return c ? (void*)c->stream : NULL; }
int stocs_set_stream(stocs_ctx* c, void* hip_stream) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));   // nothing of the old stream may still be in flight
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return STOCS_OK;
}

int stocs_dev_alloc(stocs_ctx* c, int64_t bytes, void** dptr) {
    if (!c || !dptr || bytes < 0) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    STOCS_HIP_CHECK(dev_malloc(dptr, (size_t)std::max<int64_t>(bytes, 16)));
    return STOCS_OK;
}
int stocs_dev_free(stocs_ctx* c, void* dptr) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (dptr) STOCS_HIP_CHECK(hipFree(dptr));
    return STOCS_OK;
}
int stocs_dev_upload(stocs_ctx* c, void* dptr, const void* host, int64_t bytes) {
    if (!c || !dptr || !host) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    STOCS_HIP_CHECK(hipMemcpyAsync(dptr, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}
int stocs_dev_download(stocs_ctx* c, void* host, const void* dptr, int64_t bytes) {
    if (!c || !dptr || !host) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    STOCS_HIP_CHECK(hipMemcpyAsync(host, dptr, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return STOCS_OK;
}

int stocs_last_call_timing(const stocs_ctx* c, int which, const char** labels, double* ms, int cap, int* n) {
    if (!c || which < 0 || which > 3 || !n || cap < 0) return STOCS_ERR_INVALID;
    const CallTiming& t = c->timing[which];
    *n = t.n;
    for (int i = 0; i < t.n && i < cap; ++i) {
        if (labels) labels[i] = t.label[i];
        if (ms) ms[i] = t.ms[i];
    }
    return t.n > cap ? STOCS_ERR_CAPACITY : STOCS_OK;
}

// CPU-only self test of the STOCS_DEBUG_STREAMS checker (stream_audit.h) on canned two-stream sequences: 0 = the fork / join
// pattern of the library (no violation), 1 = a use on the auxiliary stream without the main -> aux edge, 2 = the main stream reads
// what the auxiliary stream wrote without the aux -> main edge, 3 = the arena is recycled while the auxiliary stream still writes,
// 4 = a write behind a read of the other stream without an edge.  Returns the number of violations found.
int stocs_debug_streams_overlap(stocs_ctx* c) {
    if (!c) return STOCS_ERR_INVALID;
    DeviceGuard dev_guard(c->device);
    if (!c->own_stream || !c->aux_stream) return 0;
    STOCS_HIP_CHECK(hipStreamSynchronize(c->own_stream));
    STOCS_HIP_CHECK(hipStreamSynchronize(c->aux_stream));
    return streams_run_side_by_side(c->own_stream, c->aux_stream);
}

int stocs_debug_stream_audit_selftest(int scenario, char* first_msg, int cap) {
    StreamAudit A;
    A.begin(true);
    int buf_a = 0, buf_b = 0, ev_fork = 0, ev_join = 0;
    A.use(0, &buf_a, true, "a", "produce a (main)");
    if (scenario != 1) { A.record(&ev_fork, 0); A.wait(1, &ev_fork); }
    A.use(1, &buf_a, false, "a", "consume a (aux)");
    A.use(1, &buf_b, true, "b", "produce b (aux)");
    if (scenario == 4) A.use(0, &buf_a, true, "a", "overwrite a (main)");
    if (scenario != 2 && scenario != 3) { A.record(&ev_join, 1); A.wait(0, &ev_join); }
    if (scenario != 3) A.use(0, &buf_b, false, "b", "consume b (main)");
    if (scenario == 3) { A.host_sync(0); A.retire_all("arena reset"); }
    else { A.host_sync(0); if (scenario == 0) A.retire_all("arena reset"); }
    if (first_msg && cap > 0) { first_msg[0] = 0; if (!A.violations.empty()) snprintf(first_msg, (size_t)cap, "%s", A.violations[0].c_str()); }
    return (int)A.violations.size();
}

int64_t stocs_device_alloc_count(void) { return (int64_t)__atomic_load_n(&g_dev_allocs, __ATOMIC_RELAXED); }

uint64_t stocs_pack_best(float lcp, uint32_t id) {
    // a score that is not positive never wins (stocs.cpp:987-998: strict > from 0, all-zero => no pose): its key is 0 =
    // "none", the same value best_kernel produces, so a rank whose candidates all scored 0 cannot win the all-reduce
    if (!(lcp > 0.0f)) return 0;
    uint32_t bits;
    memcpy(&bits, &lcp, 4);
    return ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - id);
}
void stocs_unpack_best(uint64_t key, float* lcp, uint32_t* id) {
    uint32_t bits = (uint32_t)(key >> 32);
    if (lcp) memcpy(lcp, &bits, 4);
    if (id) *id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
}

}  // extern "C"
