/*
 * stocs_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (single-threaded C++, float arithmetic, libm) of the StoCS hot path of
 * kuwt/model_matching: src/stocs.cpp, src/rgbd.cpp:85-154, include/super4pcs/accelerators/
 * {kdtree.h,normalset.h,normalset.hpp,bbox.h,utils.h}, include/super4pcs/pairCreationFunctor.h,
 * src/pose_clustering.cpp.  Every function cites the reference file:line it follows.
 *
 * PARITY STATUS: "parity unpinned".  The reference holds no tests, golden vectors or expected
 * outputs for this path and cannot be built here (needs Eigen, PCL,
 * OpenCV, Boost -- all absent).  This restatement is pinned only by hand-derived known-answer
 * tests (tests/test_oracle_*.py) and by internal cross-checks (kd-tree vs brute force, literal
 * 128-key std::map index vs the query-side form).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (model_matching_amd/) never links, imports or calls it.
 *
 * Plain C ABI so that Python (ctypes) can drive it.
 */
#ifndef STOCS_ORACLE_H
#define STOCS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

typedef struct orc_params {
    float distance_threshold;      /* stocs_match_one_object.cpp:8  (0.005) */
    int   ppf_tr_discretization;   /* :9  (5 mm)  */
    int   ppf_rot_discretization;  /* :10 (5 deg) */
    float plane_threshold;         /* stocs.cpp:368 (0.015) */
    float min_distance_base;       /* stocs.cpp:369 (0.01)  */
    float internal_angle_threshold;/* stocs.cpp:370 (30)    */
    int   image_width, image_height;
} orc_params;

void orc_default_params(orc_params* p);

/* ---- row 1: PPF (rgbd.cpp:85-121) ---- */
int  orc_ppf_closest_bin(int value, int discretization);
/* mode 0: atan2 evaluated in double on float->double promoted arguments (the convention of this
 * repo, see DESIGN.md "numerics"); mode 1: atan2f then float*180 then /M_PI in double. */
void orc_ppf_compute(const float* p1, const float* n1, const float* p2, const float* n2,
                     int tr, int rot, int mode, int* out4);

/* ---- rows 2/T4: PPF index ---- */
typedef struct orc_index orc_index;
orc_index* orc_index_build(const float* pos3, const float* nrm3, int n, int tr, int rot);
void orc_set_index_build_threads(int n);   /* feature evaluation only; the index does not depend on it */
void       orc_index_free(orc_index*);
/* lookup(K): pairs in insertion (lexicographic id1,id2) order; returns total count, writes <= cap */
int64_t    orc_index_lookup(const orc_index*, const int* key4, int32_t* pairs2, int64_t cap);
int        orc_index_exists(const orc_index*, const int* key4);
int64_t    orc_index_num_pairs(const orc_index*);
/* literal std::map with the 128-key insertion of rgbd.cpp:123-154 (small n only) */
typedef struct orc_index_lit orc_index_lit;
orc_index_lit* orc_index_lit_build(const float* pos3, const float* nrm3, int n, int tr, int rot);
void       orc_index_lit_free(orc_index_lit*);
int64_t    orc_index_lit_lookup(const orc_index_lit*, const int* key4, int32_t* pairs2, int64_t cap);
int64_t    orc_index_lit_num_keys(const orc_index_lit*);

/* ---- context: ctor steps stocs.hpp:46-57 (centroid_shift + kdtree_initialize) ---- */
orc_ctx* orc_ctx_create(const orc_params* prm,
                        const float* scene_pos3, const float* scene_nrm3, const float* scene_prob,
                        const int32_t* scene_pixel2, int nS,
                        const float* model_pos3, const float* model_nrm3, int nM,
                        int build_index);
void     orc_ctx_destroy(orc_ctx*);
void     orc_get_centroids(const orc_ctx*, float* scene3, float* model3);
void     orc_get_scene(const orc_ctx*, float* pos3, float* prob, float* class_prob);
void     orc_get_model(const orc_ctx*, float* pos3);
void     orc_set_edge_map(orc_ctx*, const uint8_t* edge /* h*w, png value */);
const orc_index* orc_ctx_index(const orc_ctx*);

/* ---- rows 3-7: base sampling.  The clock-seeded std::discrete_distribution of
 * stocs.cpp:133-148 is replaced (documented divergence Q6) by a seeded fixed-point draw:
 * W_i = (uint64)(w_i * 2^32); r = mulhi64(rng(seed,attempt,k), sum W); index = first i with
 * inclusive prefix > r.  See DESIGN.md. ---- */
uint64_t orc_rng(uint64_t seed, uint64_t attempt, uint64_t k);
int  orc_draw(const float* w, int n, uint64_t r64);   /* -1 when all weights are zero */
int  orc_sample_class_base(orc_ctx*, uint64_t seed, uint64_t attempt, int32_t* ids4, float* inv2);
/* `segment` of the last orc_sample_instance_base (stocs.cpp:628-638) as scene indices; returns the count */
int orc_get_segment(const orc_ctx*, int32_t* idx, int cap);
int  orc_sample_instance_base(orc_ctx*, uint64_t seed, uint64_t attempt, float dispersion,
                              int base_num, int32_t* ids4, float* inv2);
/* pass-by-pass access for kernel parity: weights after pass k (k=1..3) given fixed base points */
void orc_class_pass(orc_ctx*, int pass, const int32_t* b3, const float* w_in, float* w_out);
/* row 6/7 */
double orc_segment_distance_and_invariants(const float* p1, const float* p2, const float* q1,
                                           const float* q2, double* inv1, double* inv2);
int  orc_try_sampled_base(orc_ctx*, int32_t* ids4 /*in/out*/, float* inv2);

/* ---- rows 8-10: congruent sets ---- */
int64_t orc_find_congruent(orc_ctx*, const int32_t* ids4, float inv1, float inv2,
                           int32_t* quads4, int64_t cap);
/* the same quads in the order the loop of stocs.cpp:827-858 inserts them (what the seeded subset rule draws from) */
int64_t orc_find_congruent_seq(orc_ctx*, const int32_t* ids4, float inv1, float inv2,
                               int32_t* quads4, int64_t cap);
/* introspection for KATs */
int  orc_normalset_params(float eps_unit, int* gridDepth, int* egSize, float* cell);
int  orc_cone_samples(float cos_alpha);
int  orc_index_normal(const float* n3);
float orc_model_ratio(orc_ctx*, float* gcenter3);

/* ---- rows 11-12: rigid transform ---- */
/* returns 1 when a candidate is produced (ok && rms>=0), 0 otherwise.  T16/pose16 column-major */
int  orc_rigid_transform(orc_ctx*, const int32_t* ids4, const int32_t* quad4,
                         float* T16_centred, float* pose16_camera);

/* ---- rows 14-17: kd-tree, LCP, arg-max ---- */
int   orc_nn(orc_ctx*, const float* q3, float sqdist);
int   orc_nn_brute(orc_ctx*, const float* q3, float sqdist, int* n_ties);
float orc_lcp(orc_ctx*, const float* T16);
/* batch; nthreads>1 uses OpenMP over candidates with one query stack per thread */
void  orc_lcp_batch(orc_ctx*, const float* T16, int n, float* out, int nthreads);
/* per-model-point hit index (-1 none) and counted flag, for kernel-level parity */
void  orc_lcp_detail(orc_ctx*, const float* T16, int32_t* hit, uint8_t* counted);
/* out: the reference's float-accumulated scores; out_exact: the same matches summed in double (a checker for the
 * checker: the product returns the exact mean of the weights) */
void  orc_lcp_batch_exact(orc_ctx*, const float* T16, int n, float* out, double* out_exact, int nthreads);
int   orc_best(const float* lcp, int n, float* best_score);
/* acosf-derived predicate of stocs.cpp:1028-1032 on a raw dot product */
int   orc_normal_compatible(float dot);
/* internal-angle predicate of stocs.cpp:428-429,440 on a raw dot product: 1 = zeroed */
int   orc_internal_angle_reject(float dot, float threshold);

/* ---- row 18: greedy clustering (pose_clustering.cpp:79-121) ---- */
int   orc_greedy_clustering(const float* poses16, const float* lcp, int n,
                            float acceptable_fraction, float best_score, int maximum_pose_count,
                            float min_distance, float min_angle, const float* sym3,
                            int32_t* out_idx, int cap);
void  orc_pose_diff(const float* test16, const float* base16, const float* sym3,
                    float* rot_err, float* tr_err);

/* ---- row 19: whole run (own driver semantics, seeded) ---- */
typedef struct orc_run_result {
    int   n_bases, n_quads_total, n_candidates;
    float best_lcp; int best_index;
    float best_pose16[16];
    double t_sample_s, t_congruent_s, t_verify_s;
} orc_run_result;
int orc_run(orc_ctx*, uint64_t seed, int number_of_bases, int maximum_congruent_sets,
            orc_run_result* out);
/* the same with the caller's branch on probability_maps/edge.png taken (stocs_match_one_object.cpp:90): instance_mode != 0
 * draws every base with sample_instance_base(..., dispersion, attempt + 1) */
int orc_run_mode(orc_ctx*, uint64_t seed, int number_of_bases, int maximum_congruent_sets, int instance_mode, float dispersion,
                 orc_run_result* out);
int orc_get_candidates(orc_ctx*, float* T16, float* pose16, int32_t* base_idx, int cap);

#ifdef __cplusplus
}
#endif
#endif
