/*
 * stocs_hip.h -- C ABI of libstocs_hip.so: the MI355X (gfx950) implementation of the StoCS hot path
 * of kuwt/model_matching (congruent-set sampling + rigid-transform estimation + LCP verification).
 *
 * The reference has no plugin/FFI interface: its boundary is the C++ class stocs::stocs_estimator
 * (reference include/stocs.hpp:16-180) statically linked into stocs_single
 * (reference src/stocs_match_one_object.cpp:51-185).  Each entry point below replaces one method
 * (or a batch of calls to one method) of that class; include/stocs.hpp of THIS repo is the façade
 * with the reference's method names on top of these calls, INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on success or
 * a negative stocs_status; nothing throws across the boundary; output buffers are caller-owned with
 * explicit capacities; 4x4 matrices are 16 floats COLUMN-major (Eigen::Matrix4f::data() layout);
 * one context is bound to one HIP device and one stream, is not thread-safe, distinct contexts are
 * independent.  There is no CPU fallback: without a usable HIP device stocs_ctx_create fails with
 * STOCS_ERR_NO_DEVICE.
 */
#ifndef STOCS_HIP_H
#define STOCS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct stocs_ctx stocs_ctx;

typedef enum stocs_status {
    STOCS_OK = 0,
    STOCS_ERR_INVALID = -1,    /* bad argument */
    STOCS_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime error at init */
    STOCS_ERR_HIP = -3,        /* HIP runtime error (see stocs_last_error) */
    STOCS_ERR_CAPACITY = -4,   /* caller buffer too small; required count is returned */
    STOCS_ERR_STATE = -5,      /* call order violated (e.g. no index, no bases) */
    STOCS_ERR_NOMEM = -6
} stocs_status;

/* Parameters: the file-scope constants of reference src/stocs_match_one_object.cpp:7-17 and the
 * hard-coded thresholds of reference src/stocs.cpp:368-370. */
typedef struct stocs_params {
    float distance_threshold;        /* 0.005  (:8)  epsilon of congruent-set matching and LCP   */
    int   ppf_tr_discretization;     /* 5 mm   (:9)                                               */
    int   ppf_rot_discretization;    /* 5 deg  (:10)                                              */
    float plane_threshold;           /* 0.015  (stocs.cpp:368)                                    */
    float min_distance_base;         /* 0.01   (stocs.cpp:369)                                    */
    float internal_angle_threshold;  /* 30     (stocs.cpp:370)                                    */
    float lcp_normal_angle;          /* 30     (stocs.cpp:1032)                                   */
    int   image_width, image_height; /* 640x480 (:23-24); only used by instance-mode sampling     */
    int   number_of_bases;           /* 100    (:16)                                              */
    int   maximum_congruent_sets;    /* 200    (:17)                                              */
} stocs_params;

void stocs_default_params(stocs_params* p);
const char* stocs_last_error(void);
const char* stocs_version(void);

/* ---- construction: replaces stocs_estimator::stocs_estimator (stocs.hpp:18-61):
 * load_object_info + load_scene_info (clouds are handed over as flat arrays instead of PLY/PNG
 * files), centroid_shift (stocs.cpp:943-964), kdtree_initialize (stocs.cpp:966-980; here a brick
 * grid over the centred scene).  Normals are normalised as Point3D::set_normal does.  When
 * build_index != 0 the model PPF index (reference PPFMapType, built offline by
 * stocs::pre_process_model, stocs.cpp:63-78 + rgbd.cpp:123-154) is built on the device.
 * scene_pixel2 (row,col per point) may be NULL.  device < 0 selects the current device. ---- */
int stocs_ctx_create(const stocs_params* prm,
                     const float* scene_pos3, const float* scene_nrm3, const float* scene_prob,
                     const int32_t* scene_pixel2, int nS,
                     const float* model_pos3, const float* model_nrm3, int nM,
                     int build_index, int device, stocs_ctx** out);
int stocs_ctx_destroy(stocs_ctx* ctx);
/* A new scene (the next camera frame) against the same model: load_scene_info + centroid_shift +
 * kdtree_initialize (stocs.hpp:36-60, stocs.cpp:943-980) for the scene only.  The model clouds and the PPF
 * index are kept; bases, candidates, edge map and instance-mode segments of the old scene are dropped.
 * Equivalent to destroying the context and creating it again with the same model, minus the index build. */
int stocs_ctx_set_scene(stocs_ctx* ctx, const float* scene_pos3, const float* scene_nrm3, const float* scene_prob,
                        const int32_t* scene_pixel2, int nS);

/* getters: stocs.hpp:115-116 get_scene_centroid (+ model centroid) */
int stocs_get_centroids(const stocs_ctx* ctx, float* scene3, float* model3);
int stocs_get_sizes(const stocs_ctx* ctx, int* nS, int* nM);
/* edge map (png values, image_height*image_width bytes): presence switches the driver to
 * sample_instance_base, as the stat() of probability_maps/edge.png does (stocs_match_one_object.cpp:90) */
int stocs_set_edge_map(stocs_ctx* ctx, const uint8_t* edge);
/* start a new trial stream on the same scene: restores the class probabilities given at construction
 * (instance-mode sampling decays them in place, stocs.cpp:572-580) and clears the segmentation state,
 * bases, quads and candidates.  Equivalent to constructing a new estimator, without rebuilding the grid
 * and the index. */
int stocs_reset_trial(stocs_ctx* ctx);

/* ---- PPF index queries: replace ppf_map.find (call sites stocs.cpp:403,438,487,780,784) ---- */
int stocs_ppf_compute_host(const float* p1, const float* n1, const float* p2, const float* n2,
                           int tr, int rot, int32_t* key4);       /* rgbd.cpp:99-121 */
int stocs_index_exists(const stocs_ctx* ctx, const int32_t* key4, int* exists);
/* pairs of lookup(key) in lexicographic (id1,id2) order == the reference's insertion order */
int stocs_index_lookup(stocs_ctx* ctx, const int32_t* key4, int32_t* pairs2, int64_t cap, int64_t* n);
int stocs_index_stats(const stocs_ctx* ctx, int64_t* n_pairs, int64_t* n_buckets, int64_t* n_keys);
/* on-disk form of the index: replaces rgbd::save_ppf_map / load_ppf_map (rgbd.cpp:156-177; a Boost
 * binary archive of the std::map there, compiler/Boost-version specific) by a flat, versioned,
 * little-endian CSR file.  load requires a context created with build_index = 0 for the SAME model
 * cloud and discretisation (checked through a hash stored in the file). */
int stocs_index_save(stocs_ctx* ctx, const char* path);
int stocs_index_load(stocs_ctx* ctx, const char* path);

/* ---- base sampling: batched form of the loop stocs_match_one_object.cpp:81-101 over
 * sample_class_base (stocs.cpp:363-519) / sample_instance_base (stocs.cpp:559-751).
 * mode 0 = class, 1 = instance.  Attempt i uses the seeded draws rng(seed, first_attempt+i, k).
 * Outputs per attempt: ids (permuted by try_sampled_base, stocs.cpp:224-268), the two invariants,
 * valid flag.  Valid bases are appended to the context's base set in attempt order. ---- */
int stocs_sample_bases(stocs_ctx* ctx, int mode, uint64_t seed, int first_attempt, int n_attempts,
                       float dispersion, int32_t* base_ids4, float* inv2, int32_t* valid);
/* `segment` of the last instance-mode attempt (the out-parameter of sample_instance_base, filled at stocs.cpp:628-638):
 * indices of the scene points that survived pass 1 inside the segmentation mask, in scene order */
int stocs_get_segment(const stocs_ctx* ctx, int32_t* scene_idx, int cap, int* n);
/* the context's scene as the estimator holds it: centred positions (centroid_shift), unit normals, CURRENT class
 * probabilities (instance-mode sampling decays them, Q8) and pixels; any pointer may be NULL */
int stocs_get_scene(const stocs_ctx* ctx, float* pos3_centred, float* nrm3, float* class_prob, int32_t* pixel2);
/* inject bases (already ordered, with their invariants) -- used by the parity tests and by callers
 * that sample elsewhere; replaces the context's base set */
int stocs_set_bases(stocs_ctx* ctx, int n, const int32_t* base_ids4, const float* inv2);
int stocs_clear_bases(stocs_ctx* ctx);
int stocs_num_bases(const stocs_ctx* ctx);
/* one pass of the weight update for fixed base points (kernel-level parity with the oracle):
 * pass k in 1..3, b3 = {P1,P2,P3} scene indices, w_in/w_out host arrays of nS floats */
int stocs_class_pass(stocs_ctx* ctx, int pass, const int32_t* b3, const float* w_in, float* w_out);
/* device self-check of the float filter that sits in front of the PPF arithmetic in the pass kernels: n_pairs seeded pairs
 * of scene points keyed both ways; *n_mismatch must come back 0, *n_undecided counts the pairs the filter handed to the
 * reference's double arithmetic */
int stocs_ppf_filter_check(stocs_ctx* ctx, uint64_t seed, int64_t n_pairs, int64_t* n_tested, int64_t* n_undecided, int64_t* n_mismatch);
/* device self-check of the fixed-point weight of the seeded draws: the kernels read trunc(w * 2^32) off the bits of w; all
 * 2^32 float patterns are compared with (uint64_t)((double)w * 4294967296.0) (0 for w <= 0 / NaN, saturating);
 * *n_mismatch must come back 0 */
int stocs_weight_fix_check(stocs_ctx* ctx, int64_t* n_mismatch);
/* try_sampled_base on four scene indices (stocs.cpp:224-268) */
int stocs_try_sampled_base(stocs_ctx* ctx, int32_t* ids4_inout, float* inv2, int* valid);
/* the seeded weighted draw itself (stocs.cpp:133-148 replacement): index or -1 */
int stocs_draw(stocs_ctx* ctx, const float* w, int n, uint64_t r64, int* index);

/* ---- congruent sets: find_congruent_sets_on_model (stocs.cpp:753-869) for every base of the base
 * set at once (per-base COUNTS; quads are produced on demand); then per-base read-back:
 * stocs_get_quads -- all quads of a base, sorted as the reference's std::set orders them (stocs.cpp:860-866);
 * stocs_get_quads_at -- the quads at given ranks of the base's WALK order: by position cell of the Q pair's query
 * point, then index position of the Q pair, then index position of the P pair (index position of a model pair = its
 * quantised feature, then (id1, id2)) -- the enumeration stocs_make_transforms samples from for a base with >= max
 * quads (the reference shuffles with an unseeded generator there, so any fixed enumeration serves). ---- */
int stocs_find_congruent_all(stocs_ctx* ctx, int64_t* total_quads);
int stocs_get_quads(stocs_ctx* ctx, int base_slot, int32_t* quads4, int64_t cap, int64_t* n);
int stocs_get_quads_at(stocs_ctx* ctx, int base_slot, const int64_t* ranks, int n, int32_t* quads4);

/* one cone query of the normal set (normalset.hpp:166-214) evaluated on the host, for tests: the 343-bit direction-cell set
 * from the reference's float arithmetic alone and the one the kernels build (a cheap filtered evaluation, the
 * reference's arithmetic where the filter cannot decide; cone_cells.h); the two must be equal.  n3 = query direction. */
int stocs_cone_cells_host(const float* n3, float cos_alpha, uint32_t* exact_bits11, uint32_t* kernel_bits11, int* n_samples,
                          int* n_undecided);

/* ---- candidate transforms: the loop stocs_match_one_object.cpp:120-147 over
 * get_rigid_transform_from_congruent_pair (stocs.cpp:871-941 -> ComputeRigidTransformation :270-361):
 * at most max_per_base quads per base (all when fewer; a seeded subset otherwise). ---- */
int stocs_make_transforms(stocs_ctx* ctx, int max_per_base, uint64_t seed, int* n_candidates);
/* single (base ids, quad) -> transform; ok=0 when the reference would not append a candidate */
int stocs_rigid_transform(stocs_ctx* ctx, const int32_t* ids4, const int32_t* quad4,
                          float* T16_centred, float* pose16_camera, int* ok);
int stocs_get_candidates(stocs_ctx* ctx, float* T16_centred, float* pose16_camera, float* lcp,
                         int32_t* base_index, int cap, int* n);

/* ---- verification: compute_alignment_score_for_rigid_transform (stocs.cpp:1006-1041), batched.
 * THE METRIC KERNEL: one candidate pose verified per transform. ---- */
int stocs_score_transforms(stocs_ctx* ctx, const float* T16_centred_host, int n, float* lcp_host);
/* device-resident variant: d_T16 and d_lcp are device pointers on the context's device; the call is
 * asynchronous on the context's stream (use stocs_sync) */
int stocs_score_transforms_device(stocs_ctx* ctx, const void* d_T16, int n, void* d_lcp);
/* score + device arg-max in one call (one host round trip): key as stocs_best_device.  The arg-max is taken in the epilogue of
 * the scoring kernel (one 64-bit atomic max per candidate: order independent), not by a second kernel */
int stocs_score_best_device(stocs_ctx* ctx, const void* d_T16, int n, void* d_lcp, uint32_t id_offset, uint64_t* key);
/* the same without any host synchronisation: scores into d_lcp, the packed key of the first maximum (0: no positive score) into
 * the 8 bytes at d_key8 (device memory), everything on the context's stream -- what bench.py times per step */
int stocs_score_best_device_async(stocs_ctx* ctx, const void* d_T16, int n, void* d_lcp, uint32_t id_offset, void* d_key8);
/* per model point: index of the matched scene point (-1 none) and whether it was counted */
int stocs_lcp_detail(stocs_ctx* ctx, const float* T16_centred_host, int32_t* hit, uint8_t* counted);
/* measurement aid (bench.py's `needed_bytes_per_launch`; no reference counterpart): over n device-resident transforms, the number of
 * (candidate, model point) queries of stocs.cpp:1016-1024 that found a scene point within epsilon (*hits: the nearest-neighbour
 * records a pose really needs, 28 bytes each) and how many of those passed the normal test of :1028-1032 (*counted, may be NULL).
 * Runs the per-point detail form of the scoring kernel in chunks; synchronises. */
int stocs_lcp_hit_count(stocs_ctx* ctx, const void* d_T16, int n, int64_t* hits, int64_t* counted);
/* compute_best_transform (stocs.cpp:982-1004): score every stored candidate, arg-max with first
 * maximum winning; best_idx = -1 and best_lcp = 0 when every score is 0 */
int stocs_verify_all(stocs_ctx* ctx, float* best_lcp, int* best_idx, float* best_pose16_camera);
/* ---- trial batches: N independent StoCS trials in ONE set of launches.  The reference runs one trial per process
 * (stocs_match_one_object.cpp:81-165: 100 base attempts -> congruent sets -> <= 200 candidates per base -> best candidate);
 * BASELINE config 4 runs 64 of them.  Trial t of the batch is, bit for bit (bases, congruent sets, candidates, scores, winner),
 *     stocs_reset_trial; stocs_sample_bases(mode, seeds[t], 0, n_attempts, dispersion); stocs_find_congruent_all;
 *     stocs_make_transforms(max_per_base, seeds[t]); stocs_verify_all
 * but the trials share every launch: one sampling launch (class mode: a workgroup per attempt of every trial; instance mode: a
 * workgroup pair per trial on its own copy of the image-space state), one congruent-set pass over the concatenated base sets, one
 * transform pass, one scoring launch with a per-trial arg-max.  Batches beyond what one set of launches can key or hold are cut
 * into pieces of consecutive trials (environment STOCS_TRIALS_MAX_MB: device-memory ceiling of a piece, default 16384).
 * The context is left as stocs_reset_trial leaves it (no bases, no candidates); the batch's record stays readable through the
 * getters below until the next batch.  keep_details != 0 also keeps every trial's candidate list (tests; costs a download). ---- */
typedef struct stocs_trial_result {
    int32_t n_bases;          /* valid bases among the trial's attempts                                   */
    int32_t n_candidates;     /* candidates its stocs_make_transforms produced                            */
    int64_t n_quads;          /* congruent sets over all its bases                                        */
    float   best_lcp;         /* compute_best_transform of the trial alone (stocs.cpp:982-1004)           */
    int32_t best_index;       /* index into the trial's own candidate list; -1 (and best_lcp 0): no pose  */
    float   best_pose16[16];  /* camera frame, column-major; zeros when there is no pose                  */
} stocs_trial_result;
int stocs_run_trials(stocs_ctx* ctx, int mode, int n_trials, const uint64_t* seeds, int n_attempts, float dispersion, int max_per_base,
                     int keep_details, stocs_trial_result* out /* n_trials, may be NULL */);
/* attempts of one trial of the last batch, as stocs_sample_bases returns them (ids permuted by try_sampled_base, invariants, valid) */
int stocs_trials_get_bases(stocs_ctx* ctx, int trial, int32_t* base_ids4, float* inv2, int32_t* valid, int cap_attempts, int* n_attempts);
/* congruent sets of each valid base of the trial (what stocs_get_quads reports as *n for that base slot) */
int stocs_trials_get_quad_counts(stocs_ctx* ctx, int trial, int64_t* counts, int cap, int* n);
/* the trial's candidates as stocs_get_candidates returns them after stocs_verify_all (needs keep_details; base_index counts the
 * trial's own valid bases) */
int stocs_trials_get_candidates(stocs_ctx* ctx, int trial, float* T16_centred, float* pose16_camera, float* lcp, int32_t* base_index, int cap, int* n);

/* arg-max of n device-resident scores on the device: *key = max over i of
 * stocs_pack_best(lcp[i], id_offset + i), 0 when no score is positive (first maximum wins, as the
 * strict > of stocs.cpp:994).  Synchronises the context's stream; 8 bytes cross PCIe. */
int stocs_best_device(stocs_ctx* ctx, const void* d_lcp, int n, uint32_t id_offset, uint64_t* key);
/* order-preserving key for the cross-GPU arg-max (max wins; lowest global id wins ties); 0 = "no pose" when the score
 * is not positive (stocs.cpp:987-998), so a rank whose candidates all scored 0 never wins */
uint64_t stocs_pack_best(float lcp, uint32_t global_candidate_id);
void stocs_unpack_best(uint64_t key, float* lcp, uint32_t* global_candidate_id);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (librccl is opened lazily).  The path has a single
 * exchange: the arg-max of compute_best_transform across ranks. ---- */
typedef struct stocs_comm stocs_comm;
int stocs_comm_unique_id(void* id128);                       /* rank 0: 128-byte ncclUniqueId to hand to the others */
int stocs_comm_create(const void* id128, int nranks, int rank, int device, stocs_comm** out);
int stocs_comm_destroy(stocs_comm* comm);
/* in: this rank's packed key (0 = none) and pose; out: the global maximum and the winner's pose.
 * Global ids must satisfy id / ids_per_rank == owning rank.  Status: exercised with a one-rank communicator on a
 * one-GPU box and by world-size-2 gloo tests of the same reduction; the 2+ rank RCCL path is unmeasured so far. */
int stocs_allreduce_best(stocs_comm* comm, void* hip_stream, uint64_t* key_inout, float* pose16_inout, uint32_t ids_per_rank);

/* ---- pose post-processing: clustering::greedy_clustering (pose_clustering.cpp:79-121), host ---- */
int stocs_cluster_poses(const float* poses16, const float* lcp, int n, float acceptable_fraction,
                        float best_score, int maximum_pose_count, float min_distance, float min_angle,
                        const float* sym3, int32_t* out_idx, int cap, int* n_out);

/* ---- upstream rows, GPU implementations.  PARITY WITH THE REFERENCE IS UNPINNED:
 * their arithmetic lives in PCL / OpenCV-contrib, absent here; these are pinned against this repo's
 * numpy restatement (oracle/ingest_oracle.py).  Stand-alone calls (no context). ---- */
enum {
    /* surface normals of the depth image, rgbd.cpp:199-205 (cv::rgbd::RgbdNormals, RGBD_NORMALS_METHOD_LINEMOD on the raw
     * 16-bit depth image).  0: the published method restated -- Hinterstoisser et al., "Gradient Response Maps for Real-Time
     * Detection of Texture-Less Objects", PAMI 2012, section 2.4: least-squares depth gradient over the 8 neighbours at
     * +-5 pixels whose depth differs from the centre by less than 50 raw units, normal of the tangent plane through the three
     * back-projected points X, X(x+1), X(y+1), oriented toward the camera; integer sums, float normal.  Parity with OpenCV's
     * implementation is UNPINNED (library absent): patch, threshold and arithmetic follow the paper and the library's
     * documented defaults.  1: the least-squares plane over the 5x5 window of rounds 1-2 (tests/golden/example_*.npz hold its clouds) */
    STOCS_NORMALS_DEPTH_GRADIENT = 0,
    STOCS_NORMALS_PLANE_FIT = 1
};
typedef struct stocs_camera {
    float fx, cx, fy, cy;      /* stocs_match_one_object.cpp:20 */
    float depth_scale;         /* :21 */
    int width, height;         /* :23-24 */
    int normal_method;         /* STOCS_NORMALS_* */
} stocs_camera;
/* rgbd::load_rgbd_data_sampled (rgbd.cpp:179-281): depth + class-probability images (uint16) -> voxelised,
 * outlier-filtered, oriented scene cloud with class probability and (row, col) pixel per point */
int stocs_ingest_scene(const stocs_camera* cam, const uint16_t* depth, const uint16_t* class_prob, float voxel_size,
                       float class_threshold, int device, float* pos3, float* nrm3, float* prob, int32_t* pixel2,
                       int cap, int* n_out);
/* cloud part of stocs::pre_process_model (stocs.cpp:43-60): raw vertices -> radius normals pointing away
 * from the model origin -> voxel grid (positions and normals averaged per leaf) -> scale */
int stocs_preprocess_model(const float* raw_pos3, int n_raw, float normal_radius, float voxel_size, float model_scale,
                           int device, float* pos3, float* nrm3, int cap, int* n_out);
/* stocs_ingest_scene / stocs_preprocess_model keep their device workspace cached per calling thread and device
 * (a stream of frames does no hipMalloc / hipFree after the first one), and stocs_ingest_scene a pinned host block per calling thread
 * (40 bytes per pixel: the frame's two images go up and its cloud comes down through it, so the caller's arrays may be ordinary
 * pageable memory at no cost); this gives the calling thread's cache and block back. */
int stocs_trim(void);

/* ---- files either side of the path (host code; zlib only): what the reference's constructor and driver read and
 * write through OpenCV / PCL.  stocs_png_read: 8/16-bit grey / RGB (+alpha), non-interlaced; pixels = height rows of
 * width*channels samples of bit_depth/8 bytes, 16-bit in host byte order; pixels == NULL queries the sizes.
 * stocs_ply_read: ascii or binary_little_endian vertices, x y z (+ normal_x/nx ...); pos3 == NULL queries the count.
 * stocs_ply_write: ascii, positions multiplied by `scale` as rgbd::save_as_ply does (rgbd.cpp:35-56). ---- */
int stocs_png_read(const char* path, int* width, int* height, int* channels, int* bit_depth, void* pixels, int64_t cap_bytes);
int stocs_ply_read(const char* path, float* pos3, float* nrm3, int cap, int* n, int* has_normals);
int stocs_ply_write(const char* path, const float* pos3, const float* nrm3, int n, float scale);

/* clustering::point_to_plane_icp (pose_clustering.cpp:123-140: PCL IterativeClosestPointWithNormals, 5
 * iterations, 3.5 cm): own linearised point-to-plane ICP; T16_out maps the source cloud onto the target
 * (column-major).  Stand-alone; parity with PCL unpinned. */
int stocs_icp_point_to_plane(const float* src_pos3, int nsrc, const float* tgt_pos3, const float* tgt_nrm3, int ntgt,
                             int max_iterations, float max_correspondence_distance, int device, float* T16_out,
                             int* n_correspondences);

/* ---- tuning knobs (never change results beyond float summation order).
 * "lcp_variant": 99 = automatic (default): the scan fed from a per-wavefront LDS queue of the queries that have a list -- over
 *   index-ordered lists at cell edge epsilon (24, sparse scenes), over centre-sorted lists with triangle-inequality early exit
 *   at epsilon/2 or epsilon/4 (39, dense scenes); selectable cross-checks: 31 (the per-step cooperative scan of round 2 over the
 *   centre-sorted lists) and 0 (plain lane-per-query scan).  Every selectable kernel returns the reference's scores; any other
 *   value is STOCS_ERR_INVALID (the variants that lost their A/B runs exist in the measurement build only: make tools).
 * "device_clock": 1 = stocs_find_congruent_all records HIP events between its kernel groups and stocs_last_call_timing reports them as
 *   "device: ..." steps; 0 (default; STOCS_DEVICE_CLOCK=1 in the environment turns it on at context creation) = the host's steps only --
 *   an event between two kernels of a stream costs ~5 us of idle queue on this runtime, nine of them 45 us of a 600 us call.
 * "lcp_flat": 1 (default) = sparse scenes address a flat copy of the cell table (one look-up per query), 0 = brick look-ups.
 * "lcp_split": 1 (default) = four wavefronts share a candidate's model points, 0 = one wavefront per candidate.  Scores are
 *   accumulated as integers, so neither option changes a single bit of them.
 * "lcp_order": 0 = candidates in batch order, 1 (default) = big batches against scenes whose lists do not stay in the caches
 *   are processed in a spatial order of their translations (scores are bitwise independent of it), >= 2 = always ordered, in
 *   XCD-blocked variants of that order (k > 2: runs of k consecutive slots per XCD).
 * "lcp_cull": the patch test of the queue-fed scoring kernels.  The reference walks every model point of every candidate
 *   (stocs.cpp:1016-1035); a 64-point step of the model whose bounding sphere, under the candidate transform, is farther
 *   than epsilon from every scene point cannot add to the score and is skipped after one look-up in a distance field of
 *   the scene.  0 = off, 1 (default) = on once the field pays (1e9 candidates x model points scored against the scene so
 *   far: the field costs ~0.25 ms per scene and takes ~6 % off a launch), 2 = from the first call.  Same scores, bitwise.
 * "lcp_cull_after": that threshold of "lcp_cull" = 1 in MILLIONS of point queries (candidates x model points scored against
 *   the current scene); default 1000, 0 = from the first call.  One trial of a 640x480 frame is 1-40 million: a caller that
 *   scores a single trial per frame never fills the field, a trial batch (stocs_run_trials) or a stream of candidate batches
 *   crosses the threshold within a call or two.  When the scene BEFORE the current one crossed it, stocs_ctx_set_scene fills
 *   the new scene's field at once, on the context's auxiliary stream (a frame stream on a fixed camera will cross it again).
 * "lcp_group": lanes that verify one queued query together in the queue-fed kernels: 4 (two entries of a 128-byte list line per
 *   lane, sixteen queries per trip).  The only value of the product library (8, one entry per lane, is the form of rounds 1-3a
 *   and lives in the measurement build). ---- */
int stocs_set_option(stocs_ctx* ctx, const char* key, int value);
/* Diagnostics of that patch test (tests only; no reference counterpart).  patches4: n_patches x (centre x, y, z, radius) in the
 * centred model frame, one per 64 consecutive slots of the sorted model; perm: sorted slot -> model index (|M| entries);
 * geom8: origin x, y, z, cell edge, cap, nx, ny, nz of the scene's distance field; dist: its nx*ny*nz values (x fastest;
 * the field is filled by this call if it was not yet): distance from a cell's centre to the nearest scene point, rounded
 * down, capped.  NULL outputs are skipped; *n_patches and *n_dist are always set (0 when the context has no field). */
int stocs_get_cull_state(stocs_ctx* ctx, float* patches4, int32_t* perm, int* n_patches, float* geom8, float* dist,
                         int64_t dist_cap, int64_t* n_dist);
/* The model-side half of that test without a context or a device (host code: what stocs_ctx_create computes): the order in
 * which the scoring kernels walk the model -- perm[slot] = model index, 64 consecutive slots = one compact surface patch --
 * and the bounding sphere of every patch (centre x, y, z in the CENTRED model frame, radius), ceil(nM / 64) of them. */
int stocs_model_patch_order(const float* model_pos3, int nM, int32_t* perm, float* patches4);

/* ---- stream / timing plumbing ---- */
/* run the context's work on a caller-owned HIP stream (e.g. PyTorch's current stream, so that RCCL
 * collectives issued through torch.distributed are ordered after the kernels without a host sync);
 * NULL restores the context's own stream.  The caller keeps the stream alive. */
int stocs_set_stream(stocs_ctx* ctx, void* hip_stream);
/* asynchronous arg-max: writes the packed key (see stocs_best_device) to the 8 bytes at d_key8 on the
 * context's stream; no synchronisation */
int stocs_best_device_async(stocs_ctx* ctx, const void* d_lcp, int n, uint32_t id_offset, void* d_key8);
int   stocs_sync(stocs_ctx* ctx);
void* stocs_stream(stocs_ctx* ctx);          /* hipStream_t */
/* times `reps` back-to-back launches of the LCP kernel on n device-resident transforms with HIP
 * events on the context's stream; returns the average milliseconds per launch */
int stocs_time_score_kernel(stocs_ctx* ctx, const void* d_T16, int n, void* d_lcp, int reps,
                            float* avg_ms);
/* Debug aid: with STOCS_DEBUG_STREAMS=1 in the environment stocs_find_congruent_all and stocs_make_transforms -- the two entry
 * points that use the context's auxiliary stream -- check on the host, step by step, that every buffer used on both streams is
 * ordered by an event edge (main -> aux before its first use there, aux -> main before its next use or before its arena is
 * recycled) and return STOCS_ERR_STATE naming buffer and step otherwise.  The launches are the same with and without it.  This
 * runs the checker itself on canned sequences (no device): scenario 0 = the library's fork / join pattern (returns 0), 1-4 = one
 * missing edge each (return >= 1; first_msg receives the report). */
int stocs_debug_stream_audit_selftest(int scenario, char* first_msg, int cap);
/* Do the context's two streams run side by side?  1 = a kernel on the auxiliary stream ran WHILE one on the main stream was waiting
 * for it, 0 = it did not (both streams sit on one hardware queue of the runtime, GPU_MAX_HW_QUEUES: the two-stream sections of
 * stocs_find_congruent_all / stocs_make_transforms then run one after the other -- correct, ~13 % slower at Cm), < 0 = error.
 * stocs_ctx_create asks the same question and takes another auxiliary stream (up to four candidates) until the answer is 1;
 * STOCS_NO_STREAM_PROBE=1 in the environment skips that.  The context must be idle. */
int stocs_debug_streams_overlap(stocs_ctx* ctx);
/* Diagnostics of the library's own stable radix sort of (u32 key, u32 value) pairs (csrc/sort32.hip; the sort of the congruent-set
 * phase's pair lists -- in the reference a pointer grid of per-cell vectors, include/super4pcs/accelerators/normalset.hpp:114-131):
 * sorts n host pairs by key bits [0, end_bit) on `device` (-1: the current one), which = 1 with the library's sort, 0 with rocPRIM's;
 * `reps` timed runs after one warm-up; sorted pairs and the average device time of one sort come back.  seg_off (host, n_seg + 1 ascending
 * offsets; own sort only; NULL: the whole list): every segment is sorted on its own and stays where it is -- the form the congruent-set
 * phase uses, one segment per base. */
int stocs_debug_sort_pairs(int device, const uint32_t* keys, const uint32_t* vals, int64_t n, int end_bit, int which, int reps,
                           uint32_t* keys_out, uint32_t* vals_out, float* ms_per_sort, const uint32_t* seg_off, int n_seg);
/* number of device (hipMalloc) and pinned-host (hipHostMalloc) allocations the library has made in this process so far.
 * A warm context -- one that has run a trial of the current scene -- runs further trials without allocating: the
 * difference across them is 0. */
int64_t stocs_device_alloc_count(void);
/* host wall clock, in milliseconds, of the steps of the context's LAST stocs_find_congruent_all (which = 0),
 * stocs_make_transforms (1), stocs_verify_all (2) or stocs_run_trials (3: its phases summed over the pieces of the batch): always recorded (a few clock reads per call, no synchronisation of its
 * own), so that a call that stalls -- tens of milliseconds instead of one -- names the step it stalled in.  Steps are host
 * intervals between the call's existing synchronisation points: "wait for the device" steps hold the GPU work, the others
 * host work and runtime calls; entries whose label starts with "device:" are HIP-event times of the kernel groups that
 * wait covered.  At most 18 entries.  labels[i] point to static strings.  STOCS_ERR_CAPACITY when cap is too small (*n is set). */
int stocs_last_call_timing(const stocs_ctx* ctx, int which, const char** labels, double* ms, int cap, int* n);
/* device memory helpers so that callers without a HIP binding (ctypes) can keep inputs resident */
int stocs_dev_alloc(stocs_ctx* ctx, int64_t bytes, void** dptr);
int stocs_dev_free(stocs_ctx* ctx, void* dptr);
int stocs_dev_upload(stocs_ctx* ctx, void* dptr, const void* host, int64_t bytes);
int stocs_dev_download(stocs_ctx* ctx, void* host, const void* dptr, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
