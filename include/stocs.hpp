// stocs.hpp -- drop-in C++ façade with the method names, argument meaning and error behaviour of the
// reference's stocs::stocs_estimator (reference include/stocs.hpp:16-180), implemented on top of the
// C ABI of libstocs_hip.so (include/stocs_hip.h).  Header-only; needs no Eigen / PCL / OpenCV.
//
// Differences from the reference class, all forced by the absent third-party stack and documented in
// INTEGRATION.md:
//   * clouds arrive as flat arrays (stocs::SceneCloud / stocs::ModelCloud) instead of PLY/PNG paths;
//     the PPF index is built on the GPU at construction instead of being loaded from a Boost archive;
//   * Eigen::Matrix4f -> stocs::Mat4f (16 floats, column-major, same memory layout as
//     Eigen::Matrix4f::data()); Eigen::Vector3f -> stocs::Vec3f;
//   * draws are seeded (set_seed) instead of clock-seeded (divergence Q6);
//   * PoseCandidate objects are owned by the estimator and freed with it (the reference leaks them).
// Error convention as in the reference: bool returns, NULL best pose, no exceptions from the hot
// path; only construction throws (std::runtime_error) because a constructor cannot return a status.
#ifndef STOCS_FACADE_HPP
#define STOCS_FACADE_HPP

#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "stocs_hip.h"

namespace stocs {

using Scalar = float;  // reference point3d.hpp:13

struct Vec3f {
    float v[3];
    float operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
};

struct Mat4f {  // column-major, layout-compatible with Eigen::Matrix4f::data()
    float m[16];
    Mat4f() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.0f; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    float& operator()(int r, int c) { return m[c * 4 + r]; }
    const float* data() const { return m; }
    float* data() { return m; }
};

// reference include/point3d.hpp:116-139
struct Quadrilateral {
    std::array<int, 4> vertices;
    Quadrilateral(int v0, int v1, int v2, int v3) { vertices = {{v0, v1, v2, v3}}; }
    bool operator<(const Quadrilateral& rhs) const { return vertices < rhs.vertices; }
    bool operator==(const Quadrilateral& rhs) const { return vertices == rhs.vertices; }
    int operator[](int idx) const { return vertices[idx]; }
    int& operator[](int idx) { return vertices[idx]; }
};

// reference include/point3d.hpp:141-156
class PoseCandidate {
public:
    Mat4f transform;  // camera frame
    float lcp;
    int base_index;
    PoseCandidate(const Mat4f& t, float l, int b) : transform(t), lcp(l), base_index(b) {}
};

struct ModelCloud {
    std::vector<float> pos;  // 3 floats per point (model_search.ply of the reference)
    std::vector<float> nrm;
    int size() const { return (int)(pos.size() / 3); }
};

struct SceneCloud {  // what rgbd::load_rgbd_data_sampled produces (reference src/rgbd.cpp:179-281)
    std::vector<float> pos, nrm, class_probability;
    std::vector<int32_t> pixel;       // row, col per point (may be empty)
    std::vector<uint8_t> edge_map;    // image_height*image_width png values, empty when no edge.png
    int size() const { return (int)(pos.size() / 3); }
};

class stocs_estimator {
public:
    // argument order follows reference stocs.hpp:18-30 where the argument still exists
    stocs_estimator(const ModelCloud& model, const SceneCloud& scene, std::string debug_location, int image_width,
                    int image_height, float distance_threshold, int ppf_tr_discretization, int ppf_rot_discretization,
                    float edge_threshold, float class_threshold, int device = -1)
        : ctx_(NULL), best_lcp(0), best_index(-1), seed_(0), attempt_(0) {
        (void)edge_threshold; (void)class_threshold;  // unused by the hot path (edge_threshold "Not used")
        this->debug_location = debug_location;
        stocs_default_params(&prm_);
        prm_.distance_threshold = distance_threshold;
        prm_.ppf_tr_discretization = ppf_tr_discretization;
        prm_.ppf_rot_discretization = ppf_rot_discretization;
        prm_.image_width = image_width;
        prm_.image_height = image_height;
        const int rc = stocs_ctx_create(&prm_, scene.pos.data(), scene.nrm.data(), scene.class_probability.data(),
                                        scene.pixel.empty() ? NULL : scene.pixel.data(), scene.size(), model.pos.data(),
                                        model.nrm.data(), model.size(), 1, device, &ctx_);
        if (rc != STOCS_OK) throw std::runtime_error(std::string("stocs_ctx_create: ") + stocs_last_error());
        if (!scene.edge_map.empty()) stocs_set_edge_map(ctx_, scene.edge_map.data());
        has_edge_ = !scene.edge_map.empty();
        scene_ = &scene;
    }
    ~stocs_estimator() { stocs_ctx_destroy(ctx_); }
    stocs_estimator(const stocs_estimator&) = delete;
    stocs_estimator& operator=(const stocs_estimator&) = delete;

    void set_seed(uint64_t seed) { seed_ = seed; attempt_ = 0; }
    // not in the reference (one estimator per scene there): the next frame against the same model; the model
    // clouds and the PPF index are kept, everything derived from the old scene is dropped
    void set_scene(const SceneCloud& scene) {
        const int rc = stocs_ctx_set_scene(ctx_, scene.pos.data(), scene.nrm.data(), scene.class_probability.data(),
                                           scene.pixel.empty() ? NULL : scene.pixel.data(), scene.size());
        if (rc != STOCS_OK) throw std::runtime_error(std::string("stocs_ctx_set_scene: ") + stocs_last_error());
        if (!scene.edge_map.empty()) stocs_set_edge_map(ctx_, scene.edge_map.data());
        has_edge_ = !scene.edge_map.empty();
        scene_ = &scene;
        attempt_ = 0; best_lcp = 0; best_index = -1;
        all_transforms.clear(); all_pose_store_.clear();
    }
    bool has_edge_map() const { return has_edge_; }
    stocs_ctx* context() { return ctx_; }

    // reference stocs.hpp:80-83 / stocs.cpp:363-519
    bool sample_class_base(std::vector<int>& base_indices, float& invariant1, float& invariant2) {
        return sample_one(0, 0.0f, base_indices, invariant1, invariant2);
    }
    // reference stocs.hpp:85-91 / stocs.cpp:559-751; `segment` is not produced (unused by the caller)
    bool sample_instance_base(std::vector<int>& base_indices, float& invariant1, float& invariant2, float dispersion,
                              int base_num) {
        attempt_ = base_num - 1;
        return sample_one(1, dispersion, base_indices, invariant1, invariant2);
    }

    // reference stocs.hpp:93-96 / stocs.cpp:753-869
    bool find_congruent_sets_on_model(std::vector<int>& base_indices, float invariant1, float invariant2,
                                      std::vector<Quadrilateral>* quadrilaterals) {
        quadrilaterals->clear();
        const int32_t ids[4] = {base_indices[0], base_indices[1], base_indices[2], base_indices[3]};
        const float inv[2] = {invariant1, invariant2};
        int64_t total = 0, n = 0;
        if (stocs_set_bases(ctx_, 1, ids, inv) != STOCS_OK || stocs_find_congruent_all(ctx_, &total) != STOCS_OK) return false;
        std::vector<int32_t> q((size_t)total * 4 + 4);
        if (stocs_get_quads(ctx_, 0, q.data(), total, &n) != STOCS_OK) return false;
        for (int64_t i = 0; i < n; ++i) quadrilaterals->emplace_back(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]);
        return quadrilaterals->size() != 0;
    }

    // reference stocs.hpp:98-101 / stocs.cpp:871-941: appends (centred transform, camera-frame pose)
    bool get_rigid_transform_from_congruent_pair(std::vector<int>& base_indices, Quadrilateral& congruent_quad, int base_index) {
        const int32_t ids[4] = {base_indices[0], base_indices[1], base_indices[2], base_indices[3]};
        const int32_t q[4] = {congruent_quad[0], congruent_quad[1], congruent_quad[2], congruent_quad[3]};
        Mat4f T, P;
        int ok = 0;
        if (stocs_rigid_transform(ctx_, ids, q, T.data(), P.data(), &ok) == STOCS_OK && ok) {
            all_transforms.push_back(T);
            all_pose_store_.emplace_back(new PoseCandidate(P, 0, base_index));
        }
        return true;  // the reference always returns true (stocs.cpp:940)
    }

    // reference stocs.hpp:103-104 / stocs.cpp:1006-1041
    Scalar compute_alignment_score_for_rigid_transform(const Mat4f& mat) {
        float s = 0;
        stocs_score_transforms(ctx_, mat.data(), 1, &s);
        return s;
    }

    // reference stocs.hpp:106-107 / stocs.cpp:982-1004 (batched on the GPU; first maximum wins)
    void compute_best_transform() {
        const int n = (int)all_transforms.size();
        std::vector<float> T((size_t)n * 16), l(n);
        for (int i = 0; i < n; ++i) std::memcpy(&T[(size_t)i * 16], all_transforms[i].data(), 64);
        Scalar max_score = 0;
        int index = -1;
        if (n > 0 && stocs_score_transforms(ctx_, T.data(), n, l.data()) == STOCS_OK) {
            for (int i = 0; i < n; ++i) {
                all_pose_store_[i]->lcp = l[i];
                if (l[i] > max_score) { max_score = l[i]; index = i; }
            }
        }
        best_lcp = max_score;
        best_index = index;
    }

    Vec3f get_scene_centroid() { Vec3f c; stocs_get_centroids(ctx_, c.v, NULL); return c; }
    std::vector<PoseCandidate*> get_pose_candidates() {
        std::vector<PoseCandidate*> out;
        for (auto& p : all_pose_store_) out.push_back(p.get());
        return out;
    }
    Scalar get_best_score() { return best_lcp; }
    PoseCandidate* get_best_pose() { return best_index == -1 ? NULL : all_pose_store_[best_index].get(); }
    const std::vector<Mat4f>& get_all_transforms() const { return all_transforms; }

protected:
    bool sample_one(int mode, float dispersion, std::vector<int>& base_indices, float& invariant1, float& invariant2) {
        int32_t ids[4] = {-1, -1, -1, -1};
        float inv[2] = {0, 0};
        int32_t valid = 0;
        if (stocs_sample_bases(ctx_, mode, seed_, attempt_++, 1, dispersion, ids, inv, &valid) != STOCS_OK) return false;
        if (base_indices.size() < 4) base_indices.resize(4);
        for (int k = 0; k < 4; ++k) base_indices[k] = ids[k];
        invariant1 = inv[0];
        invariant2 = inv[1];
        return valid != 0;
    }

    stocs_ctx* ctx_;
    stocs_params prm_;
    const SceneCloud* scene_;
    bool has_edge_;
    std::vector<Mat4f> all_transforms;
    std::vector<std::unique_ptr<PoseCandidate> > all_pose_store_;
    std::string debug_location;
    Scalar best_lcp;
    int best_index;
    uint64_t seed_;
    int attempt_;
};

}  // namespace stocs

#endif
