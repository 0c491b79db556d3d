// congruent.hip -- placeholder, replaced by the congruent-set join kernels (rows 8-10) in the next commit.
#include "stocs_ctx.h"
using namespace stocs;
extern "C" {
int stocs_find_congruent_all(stocs_ctx*, int64_t*) { set_error("stocs_find_congruent_all: not implemented yet"); return STOCS_ERR_STATE; }
int stocs_get_quads(stocs_ctx*, int, int32_t*, int64_t, int64_t*) { set_error("stocs_get_quads: not implemented yet"); return STOCS_ERR_STATE; }
}
