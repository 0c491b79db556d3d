// stocs.hpp -- drop-in C++ façade: the public surface of the reference's include/stocs.hpp (+ the types of
// include/point3d.hpp and the few rgbd:: helpers its driver uses), implemented on the C ABI of libstocs_hip.so
// (include/stocs_hip.h).  Header-only; needs no Eigen / PCL / OpenCV / Boost.
//
// A caller written against the reference -- src/stocs_match_one_object.cpp:51-215, src/model_preprocess.cpp:14-39 --
// compiles against this header unchanged: same class name, same constructor argument list (reference stocs.hpp:18-30),
// same method names, argument order and error behaviour (bool returns, NULL best pose, no exceptions from the hot
// path; only construction throws because a constructor cannot return a status), same global type names
// (Point3D, Quadrilateral, PoseCandidate, PPFMapType, MatrixType, VectorType, micro).  tests/cpp/reference_call_sequence.cpp
// is that call sequence, restated.
//
// What differs, and why (INTEGRATION.md has the table):
//   * Eigen::Matrix4f -> stocs::Mat4f (16 floats, column-major = Eigen::Matrix4f::data() layout, operator()(r, c));
//     Eigen::Vector3f -> stocs::Vec3f;
//   * PPFMapType is a handle to this repo's flat index file (written by stocs::pre_process_model), not a std::map: the
//     reference's 128-fold map does not fit memory beyond toy models (3.2e9 entries at |M| = 5000); rgbd::load_ppf_map
//     fills the handle, the estimator loads the index straight onto the GPU;
//   * files: PNG (8/16-bit, non-interlaced) and PLY (ascii / binary little endian) through the library's own readers;
//   * draws are seeded (set_seed) instead of clock-seeded (divergence Q6); PoseCandidate objects are owned by the
//     estimator and freed with it (the reference leaks them);
//   * batched methods (sample_bases / find_congruent_sets_all / make_transforms) exist next to the per-call ones: they
//     are what apps/stocs_single.cpp uses -- one GPU pass over all bases instead of one round trip per base.
#ifndef STOCS_FACADE_HPP
#define STOCS_FACADE_HPP

#include <sys/stat.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "stocs_hip.h"

namespace stocs {

struct Vec3f {   // stands in for Eigen::Matrix<float, 3, 1>
    float v[3];
    Vec3f() { v[0] = v[1] = v[2] = 0.0f; }
    Vec3f(float x, float y, float z) { v[0] = x; v[1] = y; v[2] = z; }
    float operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    float& operator()(int i) { return v[i]; }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float squaredNorm() const { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }
    float norm() const { return std::sqrt(squaredNorm()); }
    Vec3f normalized() const { const float z = squaredNorm(); if (z > 0.0f) { const float s = std::sqrt(z); return Vec3f(v[0] / s, v[1] / s, v[2] / s); } return *this; }
    Vec3f operator-(const Vec3f& o) const { return Vec3f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec3f operator+(const Vec3f& o) const { return Vec3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    const float* data() const { return v; }
    float* data() { return v; }
};

struct Mat4f {  // column-major, layout-compatible with Eigen::Matrix4f::data()
    float m[16];
    Mat4f() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.0f; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    float& operator()(int r, int c) { return m[c * 4 + r]; }
    const float* data() const { return m; }
    float* data() { return m; }
};

}  // namespace stocs

// ---- global names of the reference's headers (point3d.hpp:11-156, stocs.hpp:8-12, rgbd.hpp:23-26) ----
using Scalar = float;
using VectorType = stocs::Vec3f;
using MatrixType = stocs::Mat4f;
static constexpr Scalar kLargeNumber = 1e9;
using micro = std::chrono::microseconds;

class Point3D {   // reference include/point3d.hpp:11-92
public:
    using Scalar = float;
    using VectorType = stocs::Vec3f;
    Point3D(Scalar x, Scalar y, Scalar z) : pos_(x, y, z), rgb_(-1.0f, -1.0f, -1.0f), pixel_(0, 0), class_probability_(0), edge_probability_(0), current_probability_(0) {}
    explicit Point3D(const VectorType& other) : Point3D(other[0], other[1], other[2]) {}
    Point3D() : Point3D(0.0f, 0.0f, 0.0f) {}
    VectorType& pos() { return pos_; }
    const VectorType& pos() const { return pos_; }
    const VectorType& rgb() const { return rgb_; }
    const float& probability() const { return current_probability_; }
    const float& class_probability() const { return class_probability_; }
    const std::pair<int, int>& pixel() const { return pixel_; }
    const VectorType& normal() const { return normal_; }
    void set_rgb(const VectorType& rgb) { rgb_ = rgb; }
    void set_normal(const VectorType& normal) { normal_ = normal.normalized(); }
    void set_pixel(const std::pair<int, int>& pixel) { pixel_ = pixel; }
    void set_probability(float class_probability, float edge_probability) {
        class_probability_ = class_probability; edge_probability_ = edge_probability; current_probability_ = class_probability;
    }
    void update_class_probability(float decay_fraction) { class_probability_ = decay_fraction * class_probability_; }
    void update_probability(float new_probability) { current_probability_ = new_probability; }
    void reset_probability() { current_probability_ = class_probability_; }
    bool hasColor() const { return rgb_.squaredNorm() > Scalar(0.001); }
    Scalar& x() { return pos_.v[0]; }
    Scalar& y() { return pos_.v[1]; }
    Scalar& z() { return pos_.v[2]; }
    Scalar x() const { return pos_.v[0]; }
    Scalar y() const { return pos_.v[1]; }
    Scalar z() const { return pos_.v[2]; }
private:
    VectorType pos_, normal_, rgb_;
    std::pair<int, int> pixel_;
    float class_probability_, edge_probability_, current_probability_;
};

struct Quadrilateral {   // reference include/point3d.hpp:116-139
    std::array<int, 4> vertices;
    Quadrilateral(int v0, int v1, int v2, int v3) { vertices = {{v0, v1, v2, v3}}; }
    bool operator<(const Quadrilateral& rhs) const { return vertices < rhs.vertices; }
    bool operator==(const Quadrilateral& rhs) const { return vertices == rhs.vertices; }
    int operator[](int idx) const { return vertices[idx]; }
    int& operator[](int idx) { return vertices[idx]; }
};

class PoseCandidate {   // reference include/point3d.hpp:141-156
public:
    MatrixType transform;  // camera frame
    float lcp;
    int base_index;
    PoseCandidate(MatrixType transform, float lcp, float base_index) {
        this->transform = transform; this->lcp = lcp; this->base_index = (int)base_index;
    }
    ~PoseCandidate() {}
};

// The model's PPF index as callers hold it between rgbd::load_ppf_map and the estimator (reference rgbd.hpp:23:
// std::map<std::vector<int>, std::vector<std::pair<int,int>>>).  Here: where the flat index file lives.
struct PPFMapType {
    std::string location;   // empty: no file, the estimator builds the index on the GPU at construction
    int64_t n_pairs = 0;
    size_t size() const { return (size_t)n_pairs; }
    bool empty() const { return location.empty(); }
};

namespace stocs {
using ::Quadrilateral;
using ::PoseCandidate;
using ::Point3D;
using ::PPFMapType;

inline bool file_exists(const std::string& p) { struct stat b; return stat(p.c_str(), &b) == 0; }

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

// one image file -> samples (throws on failure)
template <class T>
inline void read_image(const std::string& path, int want_channels, int want_bits, int width, int height, std::vector<T>* out) {
    int w = 0, h = 0, c = 0, b = 0;
    if (stocs_png_read(path.c_str(), &w, &h, &c, &b, NULL, 0) != STOCS_OK) throw std::runtime_error(std::string("stocs_png_read: ") + stocs_last_error());
    if (c != want_channels || b != want_bits || (width > 0 && (w != width || h != height)))
        throw std::runtime_error(path + ": expected a " + std::to_string(width) + "x" + std::to_string(height) + " " + std::to_string(want_bits) + "-bit image with " +
                                 std::to_string(want_channels) + " channel(s)");
    out->resize((size_t)w * h * c);
    if (stocs_png_read(path.c_str(), &w, &h, &c, &b, out->data(), (int64_t)(out->size() * sizeof(T))) != STOCS_OK)
        throw std::runtime_error(std::string("stocs_png_read: ") + stocs_last_error());
}

inline void read_ply(const std::string& path, ModelCloud* m, bool need_normals) {
    int n = 0, hn = 0;
    if (stocs_ply_read(path.c_str(), NULL, NULL, 0, &n, &hn) != STOCS_OK) throw std::runtime_error(std::string("stocs_ply_read: ") + stocs_last_error());
    if (need_normals && !hn) throw std::runtime_error(path + ": no normals (run pre_process_model first)");
    m->pos.resize((size_t)n * 3); m->nrm.resize((size_t)n * 3);
    if (stocs_ply_read(path.c_str(), m->pos.data(), m->nrm.data(), n, &n, &hn) != STOCS_OK) throw std::runtime_error(std::string("stocs_ply_read: ") + stocs_last_error());
    if (need_normals) {   // rgbd::load_ply_model (rgbd.cpp:12-33): points without a finite normal are dropped
        size_t w = 0;
        for (int i = 0; i < n; ++i) {
            const float* q = &m->nrm[(size_t)i * 3];
            if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]))) continue;
            for (int k = 0; k < 3; ++k) { m->pos[w * 3 + k] = m->pos[(size_t)i * 3 + k]; m->nrm[w * 3 + k] = q[k]; }
            ++w;
        }
        m->pos.resize(w * 3); m->nrm.resize(w * 3);
    }
}

}  // namespace stocs

namespace rgbd {   // the helpers of reference include/rgbd.hpp a caller of the estimator touches

// rgbd.cpp:166-177: the reference deserialises a Boost archive into a std::map; here the handle records where the flat
// index file is (the estimator loads it onto the GPU) and how many pairs it holds
inline void load_ppf_map(std::string ppf_map_location, PPFMapType& ppf_map) {
    ppf_map.location.clear(); ppf_map.n_pairs = 0;
    FILE* f = fopen(ppf_map_location.c_str(), "rb");
    if (!f) return;
    char magic[8];
    int32_t hdr[6];
    int64_t counts[2];
    if (fread(magic, 1, 8, f) == 8 && !memcmp(magic, "STOCSIX1", 8) && fread(hdr, 4, 6, f) == 6 && fread(counts, 8, 2, f) == 2) {
        ppf_map.location = ppf_map_location;
        ppf_map.n_pairs = counts[1];
    }
    fclose(f);
}

// rgbd.cpp:35-56
inline void save_as_ply(std::string location, std::vector<Point3D>& point3d, float scale) {
    std::vector<float> p(point3d.size() * 3), n(point3d.size() * 3);
    for (size_t i = 0; i < point3d.size(); ++i)
        for (int k = 0; k < 3; ++k) { p[3 * i + k] = point3d[i].pos()[k]; n[3 * i + k] = point3d[i].normal()[k]; }
    (void)stocs_ply_write(location.c_str(), p.data(), n.data(), (int)point3d.size(), scale);
}

// rgbd.cpp:58-70
inline void transform_pointset(std::vector<Point3D>& input, std::vector<Point3D>& output, MatrixType& transform) {
    for (size_t i = 0; i < input.size(); ++i) {
        const VectorType& p = input[i].pos();
        const MatrixType& t = transform;
        output.push_back(Point3D(((t(0, 0) * p[0] + t(0, 1) * p[1]) + t(0, 2) * p[2]) + t(0, 3), ((t(1, 0) * p[0] + t(1, 1) * p[1]) + t(1, 2) * p[2]) + t(1, 3),
                                 ((t(2, 0) * p[0] + t(2, 1) * p[1]) + t(2, 2) * p[2]) + t(2, 3)));
    }
}

}  // namespace rgbd

namespace stocs {

class stocs_estimator {
public:
    // reference stocs.hpp:18-61, argument for argument
    stocs_estimator(std::string model_location, PPFMapType& ppf_map_preloaded, std::string rgb_location, std::string depth_location,
                    std::string class_probability_map_location, std::string edge_probability_map_location, std::string debug_location,
                    std::vector<float> camera_intrinsics, int image_width, int image_height, float read_depth_scale, float write_depth_scale,
                    float voxel_size, float distance_threshold, int ppf_tr_discretization, int ppf_rot_discretization, float edge_threshold,
                    float class_threshold) {
        reset_members(debug_location, image_width, image_height, distance_threshold, ppf_tr_discretization, ppf_rot_discretization, edge_threshold, class_threshold);
        load_object_info(model_location, ppf_map_preloaded);
        load_scene_info(rgb_location, depth_location, class_probability_map_location, edge_probability_map_location, camera_intrinsics, read_depth_scale,
                        write_depth_scale, voxel_size, debug_location + "/sampled_scene.ply");
        create_context(-1);
        centroid_shift();       // both happen inside stocs_ctx_create; kept for callers that spell them out
        kdtree_initialize();
    }
    // the same with the images already in memory (row-major, image_height x image_width; edge may be NULL)
    stocs_estimator(const ModelCloud& model, PPFMapType& ppf_map_preloaded, const uint16_t* depth, const uint16_t* class_probability_map,
                    const uint8_t* edge_probability_map, std::string debug_location, std::vector<float> camera_intrinsics, int image_width,
                    int image_height, float read_depth_scale, float write_depth_scale, float voxel_size, float distance_threshold,
                    int ppf_tr_discretization, int ppf_rot_discretization, float edge_threshold, float class_threshold, int device = -1) {
        reset_members(debug_location, image_width, image_height, distance_threshold, ppf_tr_discretization, ppf_rot_discretization, edge_threshold, class_threshold);
        model_ = model;
        ppf_location_ = ppf_map_preloaded.location;
        ingest(depth, class_probability_map, edge_probability_map, camera_intrinsics, read_depth_scale, write_depth_scale, voxel_size, std::string());
        create_context(device);
    }
    // clouds already sampled (tests, callers with their own ingest); the index is built on the GPU
    stocs_estimator(const ModelCloud& model, const SceneCloud& scene, std::string debug_location, int image_width, int image_height,
                    float distance_threshold, int ppf_tr_discretization, int ppf_rot_discretization, float edge_threshold, float class_threshold,
                    int device = -1) {
        reset_members(debug_location, image_width, image_height, distance_threshold, ppf_tr_discretization, ppf_rot_discretization, edge_threshold, class_threshold);
        model_ = model;
        scene_ = scene;
        create_context(device);
    }
    ~stocs_estimator() { stocs_ctx_destroy(ctx_); }
    stocs_estimator(const stocs_estimator&) = delete;
    stocs_estimator& operator=(const stocs_estimator&) = delete;

    // reference stocs.hpp:65-67 / stocs.cpp:86-97: model_search.ply + the preloaded index handle
    void load_object_info(std::string model_location, PPFMapType& ppf_map_preloaded) {
        read_ply(model_location, &model_, true);
        ppf_location_ = ppf_map_preloaded.location;
        std::cout << "|M| = " << model_.size() << ",  |map(M)| = " << ppf_map_preloaded.size() << std::endl;
    }
    // reference stocs.hpp:69-78 / stocs.cpp:99-131 -> rgbd::load_rgbd_data_sampled (rgbd.cpp:179-281), on the GPU.  The
    // rgb image only colours the reference's debug clouds and is not read.
    void load_scene_info(std::string rgb_location, std::string depth_location, std::string class_probability_map_location,
                         std::string edge_probability_map_location, std::vector<float> camera_intrinsics, float read_depth_scale, float write_depth_scale,
                         float voxel_size, std::string dst_scene_location) {
        (void)rgb_location;
        std::vector<uint16_t> depth, prob;
        std::vector<uint8_t> edge;
        read_image(depth_location, 1, 16, image_width, image_height, &depth);
        read_image(class_probability_map_location, 1, 16, image_width, image_height, &prob);
        const bool has_edge = file_exists(edge_probability_map_location);   // stat(), stocs.cpp:115
        if (has_edge) read_image(edge_probability_map_location, 1, 8, image_width, image_height, &edge);
        ingest(depth.data(), prob.data(), has_edge ? edge.data() : NULL, camera_intrinsics, read_depth_scale, write_depth_scale, voxel_size, dst_scene_location);
    }

    void set_seed(uint64_t seed) { seed_ = seed; attempt_ = 0; la_n_ = 0; la_in_ctx_ = false; }
    // not in the reference (one estimator per scene there): the next frame against the same model; the model
    // clouds and the PPF index are kept, everything derived from the old scene is dropped
    void set_scene(const SceneCloud& scene) {
        scene_ = scene;
        const int rc = stocs_ctx_set_scene(ctx_, scene_.pos.data(), scene_.nrm.data(), scene_.class_probability.data(),
                                           scene_.pixel.empty() ? NULL : scene_.pixel.data(), scene_.size());
        if (rc != STOCS_OK) throw std::runtime_error(std::string("stocs_ctx_set_scene: ") + stocs_last_error());
        if (!scene_.edge_map.empty()) stocs_set_edge_map(ctx_, scene_.edge_map.data());
        attempt_ = 0; best_lcp = 0; best_index = -1; la_n_ = 0; la_in_ctx_ = false;
        all_transforms.clear(); all_pose.clear(); all_pose_store_.clear(); batched_ = false;
    }
    bool has_edge_map() const { return !scene_.edge_map.empty(); }
    stocs_ctx* context() { return ctx_; }
    int scene_size() const { return scene_.size(); }
    int model_size() const { return model_.size(); }

    // reference stocs.hpp:80-83 / stocs.cpp:363-519
    bool sample_class_base(std::vector<int>& base_indices, float& invariant1, float& invariant2) {
        return sample_one(0, 0.0f, base_indices, invariant1, invariant2);
    }
    // reference stocs.hpp:85-91 / stocs.cpp:559-751; `segment` receives the survivors of pass 1 inside the mask (:628-638)
    bool sample_instance_base(std::vector<int>& base_indices, float& invariant1, float& invariant2, std::vector<Point3D>& segment, float dispersion,
                              int base_num) {
        attempt_ = base_num - 1;
        const bool ok = sample_one(1, dispersion, base_indices, invariant1, invariant2);
        int n = 0;
        if (stocs_get_segment(ctx_, NULL, 0, &n) == STOCS_OK && n > 0) {
            std::vector<int32_t> idx((size_t)n);
            stocs_get_segment(ctx_, idx.data(), n, &n);
            const std::vector<Point3D>& sc = scene_points();
            for (int i = 0; i < n; ++i) segment.push_back(sc[(size_t)idx[i]]);
        }
        return ok;
    }

    // reference stocs.hpp:93-96 / stocs.cpp:753-869
    bool find_congruent_sets_on_model(std::vector<int>& base_indices, float invariant1, float invariant2, std::vector<Quadrilateral>* quadrilaterals) {
        quadrilaterals->clear();
        const int32_t ids[4] = {base_indices[0], base_indices[1], base_indices[2], base_indices[3]};
        const float inv[2] = {invariant1, invariant2};
        int64_t total = 0, n = 0;
        int slot = lookahead_slot(ids, inv);
        if (slot >= 0) {                // one of the bases the facade sampled ahead: one search for all of them, then only read-backs
            if (!la_congruent_done_) {
                if (stocs_find_congruent_all(ctx_, &total) != STOCS_OK) return false;
                la_congruent_done_ = true;
            }
            if (stocs_get_quads(ctx_, slot, NULL, 0, &total) != STOCS_OK) return false;
        } else {
            la_in_ctx_ = false;
            slot = 0;
            if (stocs_set_bases(ctx_, 1, ids, inv) != STOCS_OK || stocs_find_congruent_all(ctx_, &total) != STOCS_OK) return false;
        }
        std::vector<int32_t> q((size_t)total * 4 + 4);
        if (stocs_get_quads(ctx_, slot, q.data(), total, &n) != STOCS_OK) return false;
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
            if (batched_) { all_transforms.clear(); all_pose.clear(); all_pose_store_.clear(); batched_ = false; }
            all_transforms.push_back(T);
            all_pose_store_.emplace_back(new PoseCandidate(P, 0, (float)base_index));
            all_pose.push_back(all_pose_store_.back().get());
        }
        return true;  // the reference always returns true (stocs.cpp:940)
    }

    // reference stocs.hpp:103-104 / stocs.cpp:1006-1041
    Scalar compute_alignment_score_for_rigid_transform(const MatrixType& mat) {
        float s = 0;
        stocs_score_transforms(ctx_, mat.data(), 1, &s);
        return s;
    }

    // reference stocs.hpp:106-107 / stocs.cpp:982-1004 (batched on the GPU; first maximum wins)
    void compute_best_transform() {
        if (batched_) {   // candidates live on the device (make_transforms): score + arg-max there, one round trip
            float pose[16];
            std::cout << "Transforms to verify: " << n_batched_ << std::endl;   // stocs.cpp:985
            if (stocs_verify_all(ctx_, &best_lcp, &best_index, pose) != STOCS_OK) {
                // the reference's error convention: text on stdout, "no pose" state, no exception from the hot path
                std::cout << "compute_best_transform failed: " << stocs_last_error() << std::endl;
                best_lcp = 0;
                best_index = -1;
            }
            std::cout << "best index: " << best_index << ", maximum score: " << best_lcp << std::endl;   // :1003
            fetched_ = false;
            return;
        }
        const int n = (int)all_transforms.size();
        std::cout << "Transforms to verify: " << n << std::endl;
        std::vector<float> T((size_t)n * 16), l((size_t)n);
        for (int i = 0; i < n; ++i) std::memcpy(&T[(size_t)i * 16], all_transforms[(size_t)i].data(), 64);
        Scalar max_score = 0;
        int index = -1;
        if (n > 0 && stocs_score_transforms(ctx_, T.data(), n, l.data()) == STOCS_OK) {
            for (int i = 0; i < n; ++i) {
                all_pose[(size_t)i]->lcp = l[(size_t)i];
                if (l[(size_t)i] > max_score) { max_score = l[(size_t)i]; index = i; }
            }
        }
        best_lcp = max_score;
        best_index = index;
        std::cout << "best index: " << best_index << ", maximum score: " << best_lcp << std::endl;
    }

    // reference stocs.hpp:109-113: run by the constructor there; here both are part of stocs_ctx_create (sequential f32
    // centroid sums, scene grid on the GPU), so these are idempotent no-ops kept for callers that name them
    void kdtree_initialize() {}
    void centroid_shift() {}

    VectorType get_scene_centroid() { Vec3f c; stocs_get_centroids(ctx_, c.v, NULL); return c; }
    std::vector<PoseCandidate*> get_pose_candidates() { fetch_batched(); return all_pose; }
    Scalar get_best_score() { return best_lcp; }
    int get_best_index() const { return best_index; }   // not in the reference: index of get_best_pose() in get_pose_candidates(), -1 if none
    PoseCandidate* get_best_pose() {
        if (best_index == -1) return NULL;
        fetch_batched();
        return all_pose[(size_t)best_index];
    }
    const std::vector<MatrixType>& get_all_transforms() { fetch_batched(); return all_transforms; }

    // reference stocs.hpp:136-149: the model under the best (centred-frame) transform and the scene, as PLY files
    void visualize_best_pose() {
        if (best_index == -1) return;
        fetch_batched();
        std::vector<Point3D> point3d_model_pose, model_pts = model_points_centred();
        rgbd::transform_pointset(model_pts, point3d_model_pose, all_transforms[(size_t)best_index]);
        rgbd::save_as_ply(debug_location + "/best_pose.ply", point3d_model_pose, 1);
        std::vector<Point3D> sc = scene_points();
        rgbd::save_as_ply(debug_location + "/scene.ply", sc, 1);
    }

    // ---- batched forms of the caller's loops (stocs_match_one_object.cpp:81-147), one GPU pass each ----
    // n attempts of sample_class_base / sample_instance_base (by the presence of the edge map, :90); valid bases are kept
    int sample_bases(int n_attempts, float dispersion) {
        base_ids_.assign((size_t)n_attempts * 4, -1); base_inv_.assign((size_t)n_attempts * 2, 0.0f); base_valid_.assign((size_t)n_attempts, 0);
        la_n_ = 0; la_in_ctx_ = false;
        if (stocs_clear_bases(ctx_) != STOCS_OK) return 0;
        if (stocs_sample_bases(ctx_, has_edge_map() ? 1 : 0, seed_, 0, n_attempts, dispersion, base_ids_.data(), base_inv_.data(), base_valid_.data()) != STOCS_OK) return 0;
        return stocs_num_bases(ctx_);
    }
    // find_congruent_sets_on_model for every sampled base; returns the total number of congruent sets
    long long find_congruent_sets_all() {
        int64_t total = 0;
        return stocs_find_congruent_all(ctx_, &total) == STOCS_OK ? (long long)total : -1;
    }
    // the <= maximum_congruent_sets loop (:120-147) with the library's seeded subset rule; returns the candidate count
    int make_transforms(int maximum_congruent_sets) {
        int n = 0;
        if (stocs_make_transforms(ctx_, maximum_congruent_sets, seed_, &n) != STOCS_OK) return -1;
        batched_ = true; fetched_ = false; n_batched_ = n;
        return n;
    }

    // N independent runs of that whole loop (base attempts -> congruent sets -> <= maximum_congruent_sets transforms per base ->
    // compute_best_transform) with seeds first_seed, first_seed + 1, ... in ONE set of GPU launches (stocs_run_trials): what
    // BASELINE config 4 ("64 parallel StoCS trials") runs.  Trial t equals, bit for bit, the run a fresh estimator with
    // set_seed(first_seed + t) gives through the three calls above + compute_best_transform.  The best trial's pose becomes the
    // estimator's best pose (get_best_score / get_best_pose); returns the index of that trial, -1 when no trial found a pose.
    struct TrialResult { int n_bases, n_candidates; long long n_congruent_sets; float best_lcp; int best_index; MatrixType best_pose; };
    int run_trials(int n_trials, uint64_t first_seed, int number_of_bases, int maximum_congruent_sets, float dispersion, std::vector<TrialResult>* results = NULL) {
        std::vector<uint64_t> seeds((size_t)std::max(n_trials, 0));
        for (int t = 0; t < n_trials; ++t) seeds[(size_t)t] = first_seed + (uint64_t)t;
        std::vector<stocs_trial_result> r((size_t)std::max(n_trials, 1));
        la_n_ = 0; la_in_ctx_ = false; batched_ = false; fetched_ = false; n_batched_ = 0;
        if (stocs_run_trials(ctx_, has_edge_map() ? 1 : 0, n_trials, seeds.data(), number_of_bases, dispersion, maximum_congruent_sets, 0, r.data()) != STOCS_OK) return -1;
        int best = -1;
        if (results) results->clear();
        for (int t = 0; t < n_trials; ++t) {
            if (r[(size_t)t].best_index >= 0 && (best < 0 || r[(size_t)t].best_lcp > r[(size_t)best].best_lcp)) best = t;   // strict >: the first best trial wins
            if (results) {
                TrialResult x;
                x.n_bases = r[(size_t)t].n_bases; x.n_candidates = r[(size_t)t].n_candidates; x.n_congruent_sets = (long long)r[(size_t)t].n_quads;
                x.best_lcp = r[(size_t)t].best_lcp; x.best_index = r[(size_t)t].best_index;
                memcpy(x.best_pose.data(), r[(size_t)t].best_pose16, 64);
                results->push_back(x);
            }
        }
        trial_best_.reset();
        best_lcp = 0; best_index = -1;
        if (best >= 0) {
            MatrixType m;
            memcpy(m.data(), r[(size_t)best].best_pose16, 64);
            trial_best_.reset(new PoseCandidate(m, r[(size_t)best].best_lcp, -1.0f));
            best_lcp = r[(size_t)best].best_lcp;
        }
        return best;
    }
    // the winner of the last run_trials (camera frame), NULL when no trial found a pose; owned by the estimator
    PoseCandidate* get_best_trial_pose() const { return trial_best_.get(); }

protected:
    std::unique_ptr<PoseCandidate> trial_best_;
    void reset_members(const std::string& dbg, int w, int h, float dist, int tr, int rot, float edge_thr, float class_thr) {
        ctx_ = NULL; best_lcp = 0; best_index = -1; seed_ = 0; attempt_ = 0; batched_ = false; fetched_ = false; n_batched_ = 0;
        la_n_ = 0; la_first_ = 0; la_seed_ = 0; la_cursor_ = 0; la_in_ctx_ = false; la_congruent_done_ = false; la_mode_ = 0; la_nvalid_ = 0;
        debug_location = dbg; image_width = w; image_height = h; distance_threshold = dist; ppf_tr_discretization = tr;
        ppf_rot_discretization = rot; edge_threshold = edge_thr; class_threshold = class_thr;
    }
    void ingest(const uint16_t* depth, const uint16_t* prob, const uint8_t* edge, const std::vector<float>& K, float read_depth_scale, float write_depth_scale,
                float voxel_size, const std::string& dst_scene_location) {
        if (K.size() < 4) throw std::runtime_error("camera_intrinsics must hold {fx, cx, fy, cy}");
        stocs_camera cam;
        cam.fx = K[0]; cam.cx = K[1]; cam.fy = K[2]; cam.cy = K[3]; cam.depth_scale = read_depth_scale; cam.width = image_width; cam.height = image_height;
        cam.normal_method = STOCS_NORMALS_DEPTH_GRADIENT;   // rgbd.cpp:203: RGBD_NORMALS_METHOD_LINEMOD
        const int cap = image_width * image_height;
        scene_.pos.resize((size_t)cap * 3); scene_.nrm.resize((size_t)cap * 3); scene_.class_probability.resize((size_t)cap); scene_.pixel.resize((size_t)cap * 2);
        int n = 0;
        if (stocs_ingest_scene(&cam, depth, prob, voxel_size, class_threshold, -1, scene_.pos.data(), scene_.nrm.data(), scene_.class_probability.data(),
                               scene_.pixel.data(), cap, &n) != STOCS_OK)
            throw std::runtime_error(std::string("stocs_ingest_scene: ") + stocs_last_error());
        scene_.pos.resize((size_t)n * 3); scene_.nrm.resize((size_t)n * 3); scene_.class_probability.resize((size_t)n); scene_.pixel.resize((size_t)n * 2);
        scene_.edge_map.clear();
        if (edge) scene_.edge_map.assign(edge, edge + (size_t)cap);
        if (!dst_scene_location.empty())   // rgbd::save_as_ply(dst_scene_location, point3d_scene, write_depth_scale), stocs.cpp:130
            (void)stocs_ply_write(dst_scene_location.c_str(), scene_.pos.data(), scene_.nrm.data(), n, write_depth_scale);
    }
    void create_context(int device) {
        stocs_default_params(&prm_);
        prm_.distance_threshold = distance_threshold;
        prm_.ppf_tr_discretization = ppf_tr_discretization;
        prm_.ppf_rot_discretization = ppf_rot_discretization;
        prm_.image_width = image_width;
        prm_.image_height = image_height;
        const bool load = !ppf_location_.empty();
        int rc = stocs_ctx_create(&prm_, scene_.pos.data(), scene_.nrm.data(), scene_.class_probability.data(), scene_.pixel.empty() ? NULL : scene_.pixel.data(),
                                  scene_.size(), model_.pos.data(), model_.nrm.data(), model_.size(), load ? 0 : 1, device, &ctx_);
        if (rc != STOCS_OK) throw std::runtime_error(std::string("stocs_ctx_create: ") + stocs_last_error());
        if (load && (rc = stocs_index_load(ctx_, ppf_location_.c_str())) != STOCS_OK) {
            const std::string msg = std::string("stocs_index_load: ") + stocs_last_error();
            stocs_ctx_destroy(ctx_); ctx_ = NULL;
            throw std::runtime_error(msg);
        }
        if (!scene_.edge_map.empty()) stocs_set_edge_map(ctx_, scene_.edge_map.data());
        std::cout << "|S|: " << scene_.size() << std::endl;   // stocs.cpp:970
    }
    // One attempt of the reference's one-per-call loop.  Class mode: the attempts do not depend on each other (every base
    // starts from the prior, stocs.cpp:372-381; attempt a is seeded by (seed, a)), so the facade draws a block of them in one
    // GPU pass and serves the calls from the block -- the results are those of one call per attempt.  Instance mode is
    // sequential state (and decays the class probabilities the LCP reads, Q8): one attempt per call, nothing ahead of the caller.
    bool sample_one(int mode, float dispersion, std::vector<int>& base_indices, float& invariant1, float& invariant2) {
        int32_t ids[4] = {-1, -1, -1, -1};
        float inv[2] = {0, 0};
        int32_t valid = 0;
        if (mode == 0) {
            if (!(la_mode_ == 0 && la_n_ > 0 && la_seed_ == seed_ && attempt_ >= la_first_ && attempt_ < la_first_ + la_n_)) {
                la_n_ = 0; la_in_ctx_ = false; la_congruent_done_ = false;
                la_ids_.assign((size_t)kLookahead * 4, -1); la_inv_.assign((size_t)kLookahead * 2, 0.0f); la_valid_.assign((size_t)kLookahead, 0);
                if (stocs_clear_bases(ctx_) != STOCS_OK ||
                    stocs_sample_bases(ctx_, 0, seed_, attempt_, kLookahead, 0.0f, la_ids_.data(), la_inv_.data(), la_valid_.data()) != STOCS_OK)
                    return false;
                la_first_ = attempt_; la_n_ = kLookahead; la_seed_ = seed_; la_in_ctx_ = true; la_cursor_ = 0; la_mode_ = 0;
                la_slot_of_.assign((size_t)kLookahead, -1);                  // base slot in the context of every valid attempt
                int slot = 0;
                for (int k = 0; k < kLookahead; ++k) if (la_valid_[(size_t)k]) la_slot_of_[(size_t)k] = slot++;
            }
            const size_t k = (size_t)(attempt_ - la_first_);
            for (int j = 0; j < 4; ++j) ids[j] = la_ids_[4 * k + (size_t)j];
            inv[0] = la_inv_[2 * k]; inv[1] = la_inv_[2 * k + 1];
            valid = la_valid_[k];
            ++attempt_;
        } else {
            // nothing is drawn ahead; but the context keeps the valid bases in attempt order, so the facade remembers which
            // slot each one got -- find_congruent_sets_on_model then searches once for all of them
            if (attempt_ == 0 || la_mode_ != 1) {
                la_ids_.clear(); la_inv_.clear(); la_valid_.clear(); la_slot_of_.clear();
                la_n_ = 0; la_nvalid_ = 0; la_cursor_ = 0; la_mode_ = 1;
                la_in_ctx_ = stocs_clear_bases(ctx_) == STOCS_OK;
            }
            if (stocs_sample_bases(ctx_, mode, seed_, attempt_++, 1, dispersion, ids, inv, &valid) != STOCS_OK) { la_in_ctx_ = false; return false; }
            la_ids_.insert(la_ids_.end(), ids, ids + 4); la_inv_.insert(la_inv_.end(), inv, inv + 2); la_valid_.push_back(valid);
            la_slot_of_.push_back(valid ? la_nvalid_++ : -1);
            ++la_n_;
            la_congruent_done_ = false;                                      // the base set grew: an earlier search does not cover it
        }
        if (base_indices.size() < 4) base_indices.resize(4);
        for (int k = 0; k < 4; ++k) base_indices[(size_t)k] = ids[k];
        invariant1 = inv[0];
        invariant2 = inv[1];
        return valid != 0;
    }
    // slot in the context's base set of a base the look-ahead block sampled (so that its congruent sets come out of ONE
    // search over all the block's bases), or -1
    int lookahead_slot(const int32_t* ids, const float* inv) {
        if (!la_in_ctx_ || la_n_ <= 0) return -1;
        for (int step = 0; step < la_n_; ++step) {                           // the caller walks the bases in order: found at the cursor
            const size_t k = (size_t)((la_cursor_ + step) % la_n_);
            if (la_slot_of_[k] < 0) continue;
            if (std::memcmp(&la_ids_[4 * k], ids, 16) == 0 && std::memcmp(&la_inv_[2 * k], inv, 8) == 0) { la_cursor_ = (int)k; return la_slot_of_[k]; }
        }
        return -1;
    }
    // candidates of make_transforms -> all_transforms / all_pose, on demand
    void fetch_batched() {
        if (!batched_ || fetched_) return;
        int n = 0;
        stocs_get_candidates(ctx_, NULL, NULL, NULL, NULL, 0, &n);
        std::vector<float> T((size_t)n * 16), P((size_t)n * 16), l((size_t)n);
        std::vector<int32_t> b((size_t)n);
        if (n) stocs_get_candidates(ctx_, T.data(), P.data(), l.data(), b.data(), n, &n);
        all_transforms.assign((size_t)n, Mat4f()); all_pose.clear(); all_pose_store_.clear();
        for (int i = 0; i < n; ++i) {
            std::memcpy(all_transforms[(size_t)i].data(), &T[(size_t)i * 16], 64);
            Mat4f pm; std::memcpy(pm.data(), &P[(size_t)i * 16], 64);
            all_pose_store_.emplace_back(new PoseCandidate(pm, l[(size_t)i], (float)b[(size_t)i]));
            all_pose.push_back(all_pose_store_.back().get());
        }
        fetched_ = true;
    }
    // the estimator's clouds as the reference holds them after centroid_shift (stocs.cpp:943-964)
    std::vector<Point3D> scene_points() {
        const int n = scene_.size();
        std::vector<float> p((size_t)n * 3), nr((size_t)n * 3), pr((size_t)n);
        std::vector<int32_t> px((size_t)n * 2);
        stocs_get_scene(ctx_, p.data(), nr.data(), pr.data(), px.data());
        std::vector<Point3D> out;
        out.reserve((size_t)n);
        for (int i = 0; i < n; ++i) {
            Point3D q(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
            q.set_normal(VectorType(nr[3 * i], nr[3 * i + 1], nr[3 * i + 2]));
            q.set_pixel(std::make_pair((int)px[2 * i], (int)px[2 * i + 1]));
            const float e = scene_.edge_map.empty() ? 0.0f : (float)(255.0 - scene_.edge_map[(size_t)px[2 * i] * image_width + px[2 * i + 1]]) / 255.0f;
            q.set_probability(pr[i], e);
            out.push_back(q);
        }
        return out;
    }
    std::vector<Point3D> model_points_centred() {
        float cm[3] = {0, 0, 0};
        stocs_get_centroids(ctx_, NULL, cm);
        std::vector<Point3D> out;
        for (int i = 0; i < model_.size(); ++i) {
            Point3D q(model_.pos[3 * i] - cm[0], model_.pos[3 * i + 1] - cm[1], model_.pos[3 * i + 2] - cm[2]);
            q.set_normal(VectorType(model_.nrm[3 * i], model_.nrm[3 * i + 1], model_.nrm[3 * i + 2]));
            out.push_back(q);
        }
        return out;
    }

    stocs_ctx* ctx_;
    stocs_params prm_;
    ModelCloud model_;
    SceneCloud scene_;
    std::string ppf_location_;
    std::vector<MatrixType> all_transforms;
    std::vector<PoseCandidate*> all_pose;
    std::vector<std::unique_ptr<PoseCandidate> > all_pose_store_;
    std::vector<int32_t> base_ids_, base_valid_;
    std::vector<float> base_inv_;
    std::string debug_location;
    float distance_threshold;
    int ppf_tr_discretization, ppf_rot_discretization;
    float edge_threshold, class_threshold;
    Scalar best_lcp;
    int best_index;
    int image_width, image_height;
    uint64_t seed_;
    int attempt_;
    bool batched_, fetched_;
    int n_batched_;
    // look-ahead block of class-mode attempts / the instance-mode attempts made so far (sample_one)
    enum { kLookahead = 100 };           // the reference's number_of_bases (stocs_match_one_object.cpp:16)
    std::vector<int32_t> la_ids_, la_valid_;
    std::vector<float> la_inv_;
    std::vector<int> la_slot_of_;
    uint64_t la_seed_;
    int la_first_, la_n_, la_cursor_, la_mode_, la_nvalid_;
    bool la_in_ctx_, la_congruent_done_;
};

// reference stocs.hpp:182-191 / stocs.cpp:28-84: raw model PLY -> normals (radius), flipped, voxel grid, scale -> model_search
// PLY + the PPF index of the sampled cloud, both on the GPU.  The index is built from the cloud as the estimator will read
// it back (the written file), so the two always agree bit for bit.
inline void pre_process_model(std::string src_model_location, float normal_radius, float read_depth_scale, float write_depth_scale, float voxel_size,
                              float ppf_tr_discretization, float ppf_rot_discretization, std::string dst_model_location, std::string dst_ppf_map_location) {
    ModelCloud raw, sampled;
    read_ply(src_model_location, &raw, false);
    sampled.pos.resize(raw.pos.size()); sampled.nrm.resize(raw.pos.size());
    int n = 0;
    if (stocs_preprocess_model(raw.pos.data(), raw.size(), normal_radius, voxel_size, read_depth_scale, -1, sampled.pos.data(), sampled.nrm.data(), raw.size(), &n) != STOCS_OK)
        throw std::runtime_error(std::string("stocs_preprocess_model: ") + stocs_last_error());
    std::cout << "After sampling |M|= " << n << std::endl;
    if (stocs_ply_write(dst_model_location.c_str(), sampled.pos.data(), sampled.nrm.data(), n, write_depth_scale) != STOCS_OK)
        throw std::runtime_error(std::string("stocs_ply_write: ") + stocs_last_error());
    ModelCloud written;
    read_ply(dst_model_location, &written, true);
    float max_distance = 0;   // "max distance is" (stocs.cpp:66-78), from the bounding box instead of the O(|M|^2) loop
    {
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < written.size(); ++i)
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], written.pos[3 * i + k]); hi[k] = std::max(hi[k], written.pos[3 * i + k]); }
        if (written.size()) max_distance = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    }
    std::cout << "max distance is at most: " << max_distance << std::endl;
    // a one-point scene is enough to own a context; the index depends on the model alone
    const float sp[3] = {0, 0, 1}, sn[3] = {0, 0, -1}, spr[1] = {1.0f};
    stocs_params prm;
    stocs_default_params(&prm);
    prm.ppf_tr_discretization = (int)ppf_tr_discretization;
    prm.ppf_rot_discretization = (int)ppf_rot_discretization;
    stocs_ctx* ctx = NULL;
    if (stocs_ctx_create(&prm, sp, sn, spr, NULL, 1, written.pos.data(), written.nrm.data(), written.size(), 1, -1, &ctx) != STOCS_OK)
        throw std::runtime_error(std::string("stocs_ctx_create: ") + stocs_last_error());
    const int rc = stocs_index_save(ctx, dst_ppf_map_location.c_str());
    const std::string msg = rc == STOCS_OK ? std::string() : std::string("stocs_index_save: ") + stocs_last_error();
    stocs_ctx_destroy(ctx);
    if (rc != STOCS_OK) throw std::runtime_error(msg);
}

}  // namespace stocs

#endif
