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

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
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
    STOCS_HIP_CHECK(hipMalloc(&c->d_scratch, want));
    c->scratch_bytes = want;
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
    STOCS_HIP_CHECK(hipMalloc((void**)dptr, n * sizeof(T)));
    if (h) STOCS_HIP_CHECK(hipMemcpy(*dptr, h, n * sizeof(T), hipMemcpyHostToDevice));
    return STOCS_OK;
}

// ---- scene grid build (host) -------------------------------------------------------------------
static int build_grid_div(stocs_ctx* c, int div_in) {
    SceneGrid& g = c->grid;
    const int nS = c->nS;
    STOCS_HIP_CHECK(hipStreamSynchronize(c->stream));
    { int rc0 = c->grid_mem.reset(); if (rc0) return rc0; }
    const double eps = (double)c->prm.distance_threshold;
    // cell edge = epsilon / div: a finer grid gives shorter candidate lists (closer to the epsilon ball)
    // at the price of more cells and more list copies per point
    int div = div_in;
    if (div < 1 || div > 4) div = 1;
    const double h = eps / div;
    const double r = eps * 1.001;  // safety margin >> float rounding of the device cell computation
    double mn[3] = {1e30, 1e30, 1e30}, mx[3] = {-1e30, -1e30, -1e30};
    for (int i = 0; i < nS; ++i) {
        const V3 p = c->h_spos[i];
        const double v[3] = {p.x, p.y, p.z};
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[k]); mx[k] = std::max(mx[k], v[k]); }
    }
    if (nS == 0) { for (int k = 0; k < 3; ++k) { mn[k] = 0; mx[k] = 0; } }
    const double pad = r + 2 * h;  // a query outside the grid is farther than epsilon from every point
    const double o[3] = {mn[0] - pad, mn[1] - pad, mn[2] - pad};
    int n[3];
    for (int k = 0; k < 3; ++k) n[k] = (int)floor((mx[k] + pad - o[k]) / h) + 1;
    g.ox = (float)o[0]; g.oy = (float)o[1]; g.oz = (float)o[2];
    // the device computes the cell as floor((q - o_f) * inv_h) with the float origin; re-derive the
    // exact origin the host uses from the float value so both agree
    const double of[3] = {g.ox, g.oy, g.oz};
    g.inv_h = (float)(1.0 / h);
    g.nx = n[0]; g.ny = n[1]; g.nz = n[2];
    g.nbx = (n[0] + 7) / 8; g.nby = (n[1] + 7) / 8; g.nbz = (n[2] + 7) / 8;
    const int64_t n_top = (int64_t)g.nbx * g.nby * g.nbz;
    if (n_top > (int64_t)400 * 1000 * 1000) { set_error("scene extent too large for the brick grid"); return STOCS_ERR_INVALID; }

    // (cell key, point) incidences: cell box [o + c*h, o + (c+1)*h]
    struct Inc { uint64_t key; int32_t pt; };
    std::vector<Inc> inc;
    inc.reserve((size_t)nS * 24);
    for (int i = 0; i < nS; ++i) {
        const V3 pf = c->h_spos[i];
        const double p[3] = {pf.x, pf.y, pf.z};
        int lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::max(0, (int)floor((p[k] - r - of[k]) / h));
            hi[k] = std::min(n[k] - 1, (int)floor((p[k] + r - of[k]) / h));
        }
        for (int cz = lo[2]; cz <= hi[2]; ++cz)
            for (int cy = lo[1]; cy <= hi[1]; ++cy)
                for (int cx = lo[0]; cx <= hi[0]; ++cx) {
                    const int cc[3] = {cx, cy, cz};
                    double d2 = 0;
                    for (int k = 0; k < 3; ++k) {
                        const double b0 = of[k] + cc[k] * h, b1 = b0 + h;
                        const double d = p[k] < b0 ? b0 - p[k] : (p[k] > b1 ? p[k] - b1 : 0.0);
                        d2 += d * d;
                    }
                    if (d2 > r * r) continue;
                    const uint64_t brick = ((uint64_t)(cz >> 3) * g.nby + (uint64_t)(cy >> 3)) * g.nbx + (uint64_t)(cx >> 3);
                    const uint32_t local = (uint32_t)(((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7));
                    Inc e;
                    e.key = (brick << 9) | local;
                    e.pt = i;
                    inc.push_back(e);
                }
    }
    std::sort(inc.begin(), inc.end(), [](const Inc& a, const Inc& b) { return a.key != b.key ? a.key < b.key : a.pt < b.pt; });

    // Lists are padded to multiples of 8 entries (one 128-byte line per 8 candidates) with sentinel
    // entries far away, so that 8 lanes can scan one query's list with whole-line loads.
    std::vector<int32_t> top((size_t)n_top, -1);
    std::vector<uint4> cells;
    std::vector<float4> list;
    list.reserve(inc.size() + inc.size() / 2 + 8);
    float4 sentinel; sentinel.x = sentinel.y = sentinel.z = 1.0e30f;
    { const int32_t m1 = -1; memcpy(&sentinel.w, &m1, 4); }
    int n_bricks = 0;
    std::vector<uint64_t> brick_lin;   // brick id -> linear brick index
    uint64_t cur_brick = ~0ull, cur_key = ~0ull;
    for (size_t e = 0; e < inc.size(); ++e) {
        const uint64_t brick = inc[e].key >> 9;
        const uint32_t local = (uint32_t)(inc[e].key & 511);
        if (brick != cur_brick) {
            cur_brick = brick;
            top[(size_t)brick] = n_bricks++;
            brick_lin.push_back(brick);
            uint4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
            cells.resize((size_t)n_bricks * 512, z);
        }
        if (inc[e].key != cur_key) {
            cur_key = inc[e].key;
            while (list.size() % 8) list.push_back(sentinel);
            cells[(size_t)(n_bricks - 1) * 512 + local].x = (uint32_t)list.size();
        }
        if (++cells[(size_t)(n_bricks - 1) * 512 + local].y > 65535u) { set_error("more than 65535 scene points within epsilon of one grid cell"); return STOCS_ERR_INVALID; }
        const V3 p = c->h_spos[inc[e].pt];
        float4 v; v.x = p.x; v.y = p.y; v.z = p.z;
        memcpy(&v.w, &inc[e].pt, 4);
        list.push_back(v);
    }
    while (list.size() % 8) list.push_back(sentinel);
    if (list.size() >= 0xFFFFFFF0ull) { set_error("scene grid lists too large"); return STOCS_ERR_INVALID; }
    if (getenv("STOCS_DEBUG_GRID")) {
        size_t hist[12] = {0}, ncell = 0, tot = 0, mx = 0;
        for (size_t i = 0; i < cells.size(); ++i) if (cells[i].y) {
            ncell++; tot += cells[i].y; mx = std::max<size_t>(mx, cells[i].y);
            int b = 0; while ((1u << b) < cells[i].y && b < 11) b++;
            hist[b]++;
        }
        fprintf(stderr, "[stocs grid] dims %dx%dx%d bricks %d nonempty cells %zu entries %zu (padded %zu) avg %.2f max %zu\n[stocs grid] len<=1,2,4,8,16,32,..:", g.nx, g.ny, g.nz, n_bricks, ncell, tot, list.size(), (double)tot / std::max<size_t>(ncell, 1), mx);
        for (int b = 0; b < 12; ++b) fprintf(stderr, " %zu", hist[b]);
        fprintf(stderr, "\n");
    }
    // sub-cell masks: a query can only have a neighbour within epsilon if its sub-cell's bit is set.
    // Only for cell edge = epsilon; finer grids get all-ones masks.
    if (div == 1) {
        const double hs = h / 4.0;
        for (int i = 0; i < nS; ++i) {
            const V3 pf = c->h_spos[i];
            const double p[3] = {pf.x, pf.y, pf.z};
            int lo[3], hi[3];
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::max(0, (int)floor((p[k] - r - of[k]) / hs));
                hi[k] = std::min(4 * n[k] - 1, (int)floor((p[k] + r - of[k]) / hs));
            }
            for (int sz = lo[2]; sz <= hi[2]; ++sz) {
                const double bz0 = of[2] + sz * hs, bz1 = bz0 + hs;
                const double dz = p[2] < bz0 ? bz0 - p[2] : (p[2] > bz1 ? p[2] - bz1 : 0.0);
                for (int sy = lo[1]; sy <= hi[1]; ++sy) {
                    const double by0 = of[1] + sy * hs, by1 = by0 + hs;
                    const double dy = p[1] < by0 ? by0 - p[1] : (p[1] > by1 ? p[1] - by1 : 0.0);
                    const double dyz = dy * dy + dz * dz;
                    if (dyz > r * r) continue;
                    for (int sx = lo[0]; sx <= hi[0]; ++sx) {
                        const double bx0 = of[0] + sx * hs, bx1 = bx0 + hs;
                        const double dx = p[0] < bx0 ? bx0 - p[0] : (p[0] > bx1 ? p[0] - bx1 : 0.0);
                        if (dx * dx + dyz > r * r) continue;
                        const int cx = sx >> 2, cy = sy >> 2, cz = sz >> 2;
                        const int64_t brick = ((int64_t)(cz >> 3) * g.nby + (cy >> 3)) * g.nbx + (cx >> 3);
                        const int32_t bid = top[(size_t)brick];
                        if (bid < 0) continue;  // cannot happen: the cell is within r of the point
                        uint4& cw = cells[(size_t)bid * 512 + (((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7))];
                        const int bit = ((sz & 3) << 4) | ((sy & 3) << 2) | (sx & 3);
                        if (bit < 32) cw.z |= 1u << bit; else cw.w |= 1u << (bit - 32);
                    }
                }
            }
        }
    } else {
        for (size_t i = 0; i < cells.size(); ++i) if (cells[i].y) { cells[i].z = 0xFFFFFFFFu; cells[i].w = 0xFFFFFFFFu; }
    }
    {
        size_t ncell = 0, tot = 0;
        for (size_t i = 0; i < cells.size(); ++i) if (cells[i].y) { ncell++; tot += cells[i].y; }
        g.avg_list_len = ncell ? (double)tot / (double)ncell : 0.0;
    }
    g.n_bricks = n_bricks;
    g.n_entries = (int64_t)list.size();
    g.h = (float)h;
    g.d_chunk_r = NULL;
    int rc;
    // Dense scenes (long lists): order every list by distance from its cell centre and keep, per 8-entry
    // chunk, a lower bound of that distance.  For a query q of the cell, |q - p| >= |p - c| - |q - c|, so the
    // scan may stop at the first chunk whose bound exceeds sqrt(best d^2) + |q - c|.
    if (g.avg_list_len > 16.0) {
        std::vector<float> chunk_r(list.size() / 8, 0.0f);
        std::vector<std::pair<double, float4> > tmp;
        for (int bidx = 0; bidx < n_bricks; ++bidx) {
            const uint64_t bl = brick_lin[bidx];
            const int bx = (int)(bl % g.nbx), by = (int)((bl / g.nbx) % g.nby), bz = (int)(bl / ((uint64_t)g.nbx * g.nby));
            for (int local = 0; local < 512; ++local) {
                const uint4 cw = cells[(size_t)bidx * 512 + local];
                if (!cw.y) continue;
                const int cx = bx * 8 + (local & 7), cy = by * 8 + ((local >> 3) & 7), cz = bz * 8 + (local >> 6);
                const double ccx = of[0] + (cx + 0.5) * h, ccy = of[1] + (cy + 0.5) * h, ccz = of[2] + (cz + 0.5) * h;
                tmp.clear();
                for (uint32_t k = 0; k < cw.y; ++k) {
                    const float4 e = list[cw.x + k];
                    const double dx = e.x - ccx, dy = e.y - ccy, dz = e.z - ccz;
                    tmp.push_back(std::make_pair(sqrt(dx * dx + dy * dy + dz * dz), e));
                }
                std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<double, float4>& a, const std::pair<double, float4>& b) { return a.first < b.first; });
                for (uint32_t k = 0; k < cw.y; ++k) {
                    list[cw.x + k] = tmp[k].second;
                    if ((k & 7) == 0) chunk_r[(cw.x + k) >> 3] = (float)(tmp[k].first - 2e-6);
                }
            }
        }
        if ((rc = c->grid_mem.take(std::max<size_t>(chunk_r.size(), 1) * sizeof(float), (void**)&g.d_chunk_r))) return rc;
        STOCS_HIP_CHECK(hipMemcpy(g.d_chunk_r, chunk_r.data(), chunk_r.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if ((rc = c->grid_mem.take(std::max<size_t>(top.size(), 1) * 4, (void**)&g.d_top)) ||
        (rc = c->grid_mem.take(std::max<size_t>(cells.size(), 1) * sizeof(uint4), (void**)&g.d_cells)) ||
        (rc = c->grid_mem.take(std::max<size_t>(list.size(), 8) * sizeof(float4), (void**)&g.d_list)))
        return rc;
    STOCS_HIP_CHECK(hipMemcpy(g.d_top, top.data(), top.size() * 4, hipMemcpyHostToDevice));
    STOCS_HIP_CHECK(hipMemcpy(g.d_cells, cells.data(), cells.size() * sizeof(uint4), hipMemcpyHostToDevice));
    STOCS_HIP_CHECK(hipMemcpy(g.d_list, list.data(), list.size() * sizeof(float4), hipMemcpyHostToDevice));
    return STOCS_OK;
}

// Cell edge = epsilon unless the scene is so dense that the candidate lists get long (C5: 200k points,
// 1.6 mm spacing -> 61 candidates per list): then epsilon/2 (39 per list) and, if the lists are still long,
// epsilon/4 -- every halving multiplies the list memory by ~4 and the cell count by 8, and pays as long as
// the centre-sorted early exit still has chunks to skip (C5: 50.5 -> 16.5 -> 13.2 ms).  STOCS_GRID_DIV overrides.
static void free_grid(stocs_ctx* c) {   // the grid lives in c->grid_mem, which the next build resets
    c->grid.d_top = NULL; c->grid.d_cells = NULL; c->grid.d_list = NULL; c->grid.d_chunk_r = NULL;
}

// one build at cell edge eps / div; lists longer than 16 on average get the centre-sorted layout + chunk bounds
static int build_grid_once(stocs_ctx* c, int div) {
    if (getenv("STOCS_GRID_HOST")) return build_grid_div(c, div);   // the host build, kept for A/B parity tests
    int rc = build_grid_gpu(c, div, 0);
    if (rc || c->grid.avg_list_len <= 16.0) return rc;
    free_grid(c);
    return build_grid_gpu(c, div, 1);
}

static int build_grid(stocs_ctx* c) {
    int div = c->grid_div;
    const char* e = getenv("STOCS_GRID_DIV");
    if (e) div = atoi(e);
    int rc = build_grid_once(c, div);
    if (rc || e || c->grid_div != 1) return rc;
    for (int next = 2; next <= 4; next *= 2) {
        // stop when the lists are short, or when the finer grid would not fit comfortably (entries x ~8, 16 B each)
        // (measured, profiles/r01_sweep.json: 100k points, 20 per list at eps/2 -> eps/4 is 10 % slower; 200k points, 39 per
        // list -> eps/4 is 20 % faster: the second halving needs lists well beyond the first threshold)
        if (c->grid.avg_list_len <= (next == 2 ? 16.0 : 28.0) || c->grid.n_entries * 8 >= ((int64_t)1 << 29)) break;
        const int prev = next / 2;
        free_grid(c);
        rc = build_grid_once(c, next);
        if (rc == STOCS_ERR_INVALID) {   // the finer grid does not fit the 32-bit list offsets: stay with the coarser one
            free_grid(c);
            rc = build_grid_once(c, prev);
            break;
        }
        if (rc) return rc;
    }
    if (getenv("STOCS_DEBUG_TIMING"))
        fprintf(stderr, "[stocs grid] cell edge eps/%d, %d bricks, %lld list entries (%.1f per non-empty cell)\n", (int)lround((double)c->prm.distance_threshold / c->grid.h),
                c->grid.n_bricks, (long long)c->grid.n_entries, c->grid.avg_list_len);
    return rc;
}

// Everything that depends on the scene cloud: host copies (kdtree_initialize / centroid_shift of the scene,
// stocs.cpp:943-980), device clouds, the brick grid; per-trial state is reset.  Used by stocs_ctx_create and
// stocs_ctx_set_scene (a new camera frame against the same model keeps the model clouds and the PPF index).
static int load_scene(stocs_ctx* c, const float* sp, const float* sn, const float* sprob, const int32_t* spix, int nS) {
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
    if (c->d_spos) { (void)hipFree(c->d_spos); c->d_spos = NULL; }
    if (c->d_snrmw) { (void)hipFree(c->d_snrmw); c->d_snrmw = NULL; }
    if (c->d_spix) { (void)hipFree(c->d_spix); c->d_spix = NULL; }
    free_grid(c);
    int rc = STOCS_OK;
    {
        std::vector<float4> a(std::max(nS, 1)), b(std::max(nS, 1));
        std::vector<int2> px(std::max(nS, 1));
        for (int i = 0; i < nS; ++i) {
            a[i] = make_float4(c->h_spos[i].x, c->h_spos[i].y, c->h_spos[i].z, c->h_sprob[i]);
            b[i] = make_float4(c->h_snrm[i].x, c->h_snrm[i].y, c->h_snrm[i].z, c->h_sprob[i]);
            px[i] = make_int2(c->h_spix[2 * i], c->h_spix[2 * i + 1]);
        }
        if (!rc) rc = upload(&c->d_spos, a.data(), a.size());
        if (!rc) rc = upload(&c->d_snrmw, b.data(), b.size());
        if (!rc) rc = upload(&c->d_spix, px.data(), px.size());
    }
    if (!rc) rc = build_grid(c);
    // per-trial state belongs to the old scene
    c->bases.clear(); c->quad_off.clear(); clear_candidates(c);
    c->best_lcp = 0; c->best_index = -1;
    stocs_internal_invalidate_congruent(c);
    const size_t npx = (size_t)c->prm.image_width * c->prm.image_height;
    c->has_edge = false;
    c->edge_map.assign(npx, 0);
    c->previous_segment.reset();
    c->segmentation_buffer.assign(npx, 0);
    c->seg_masks.clear();
    return rc;
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
    c->index.built = false;
    c->index.d_bucket_start = NULL; c->index.d_pairs = NULL; c->index.d_exists = NULL;
    c->cong = NULL; c->quad_id_bits = 16;
    c->d_cand = NULL; c->cand_bytes = 0; c->n_cands = 0; c->cand_cap = 0; c->cands_stale = false;
    c->d_best = NULL;
    c->best_lcp = 0; c->best_index = -1;
    c->has_edge = false;
    c->grid_div = 1;
    c->lcp_variant = -1;
    memset(&c->grid, 0, sizeof(c->grid));
    c->d_spos = c->d_snrmw = c->d_mpos = c->d_mnrm = c->d_munit = c->d_mpos_raw = c->d_mpos_s = c->d_mnrm_s = NULL;
    c->d_spix = NULL; c->d_mperm = NULL;
    c->stream = NULL; c->own_stream = NULL;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        set_error("stream/event creation failed");
        delete c;
        return STOCS_ERR_NO_DEVICE;
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

    // Morton order of the centred model for the LCP kernel (spatially coherent wavefronts)
    c->h_mperm.resize(nM);
    {
        std::vector<uint32_t> code(nM);
        const V3 ext = mk3(c->ratio, c->ratio, c->ratio);
        for (int i = 0; i < nM; ++i) {
            const V3 u = c->h_munit[i];
            auto q10 = [](float v) { int k = (int)(v * 1024.0f); return (uint32_t)(k < 0 ? 0 : (k > 1023 ? 1023 : k)); };
            code[i] = (part1by2(q10(u.z)) << 2) | (part1by2(q10(u.y)) << 1) | part1by2(q10(u.x));
        }
        (void)ext;
        std::iota(c->h_mperm.begin(), c->h_mperm.end(), 0);
        std::stable_sort(c->h_mperm.begin(), c->h_mperm.end(), [&](int a, int b) { return code[a] < code[b]; });
    }

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
        if (!rc) rc = upload(&c->d_mpos, a.data(), a.size());
        if (!rc) rc = upload(&c->d_mnrm, b.data(), b.size());
        if (!rc) rc = upload(&c->d_munit, u.data(), u.size());
        if (!rc) rc = upload(&c->d_mpos_raw, raw.data(), raw.size());
        if (!rc) rc = upload(&c->d_mpos_s, as.data(), as.size());
        if (!rc) rc = upload(&c->d_mnrm_s, bs.data(), bs.size());
        if (!rc) rc = upload(&c->d_mperm, c->h_mperm.data(), c->h_mperm.size());
    }
    if (!rc) rc = load_scene(c, sp, sn, sprob, spix, nS);
    if (!rc && build_index) rc = build_ppf_index(c);
    if (rc) { stocs_ctx_destroy(c); return rc; }
    *out = c;
    return STOCS_OK;
}

int stocs_ctx_destroy(stocs_ctx* c) {
    if (!c) return STOCS_OK;
    DeviceGuard dev_guard(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->d_spos, c->d_snrmw, c->d_spix, c->d_mpos, c->d_mnrm, c->d_munit, c->d_mpos_raw, c->d_mpos_s,
                    c->d_mnrm_s, c->d_mperm, c->index.d_bucket_start,
                    c->index.d_pairs, c->index.d_exists, c->d_scratch, c->d_best, c->d_cand};
    stocs_internal_free_congruent(c);
    c->grid_mem.destroy(); c->grid_ws.destroy();
    for (void* p : ptrs) if (p) hipFree(p);
    hipEventDestroy(c->ev0);
    hipEventDestroy(c->ev1);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
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
    STOCS_HIP_CHECK(hipMalloc(dptr, (size_t)std::max<int64_t>(bytes, 16)));
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

uint64_t stocs_pack_best(float lcp, uint32_t id) {
    uint32_t bits;
    memcpy(&bits, &lcp, 4);
    if (!(lcp > 0.0f)) bits = 0;  // scores are >= 0; NaN/negative never win
    return ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - id);
}
void stocs_unpack_best(uint64_t key, float* lcp, uint32_t* id) {
    uint32_t bits = (uint32_t)(key >> 32);
    if (lcp) memcpy(lcp, &bits, 4);
    if (id) *id = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
}

}  // extern "C"
