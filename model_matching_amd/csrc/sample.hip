// sample.hip -- placeholder, replaced by the base-sampling kernels (rows 3-7) in the next commit.
#include "stocs_ctx.h"
using namespace stocs;
#define NYI(name) do { set_error(name ": not implemented yet"); return STOCS_ERR_STATE; } while (0)
extern "C" {
int stocs_sample_bases(stocs_ctx*, int, uint64_t, int, int, float, int32_t*, float*, int32_t*) { NYI("stocs_sample_bases"); }
int stocs_set_bases(stocs_ctx* c, int n, const int32_t* ids, const float* inv) {
    if (!c || n < 0 || (n && (!ids || !inv))) return STOCS_ERR_INVALID;
    c->bases.clear(); c->quads.clear();
    for (int i = 0; i < n; ++i) { BaseRec b; for (int k = 0; k < 4; ++k) b.ids[k] = ids[4 * i + k]; b.inv1 = inv[2 * i]; b.inv2 = inv[2 * i + 1]; c->bases.push_back(b); }
    return STOCS_OK;
}
int stocs_clear_bases(stocs_ctx* c) { if (!c) return STOCS_ERR_INVALID; c->bases.clear(); c->quads.clear(); return STOCS_OK; }
int stocs_num_bases(const stocs_ctx* c) { return c ? (int)c->bases.size() : STOCS_ERR_INVALID; }
int stocs_class_pass(stocs_ctx*, int, const int32_t*, const float*, float*) { NYI("stocs_class_pass"); }
int stocs_try_sampled_base(stocs_ctx*, int32_t*, float*, int*) { NYI("stocs_try_sampled_base"); }
int stocs_draw(stocs_ctx*, const float*, int, uint64_t, int*) { NYI("stocs_draw"); }
}
