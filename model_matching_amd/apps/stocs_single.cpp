// stocs_single -- the repo's equivalent of the reference driver (reference
// src/stocs_match_one_object.cpp:51-215): same four phases, same three timed spans printed in
// microseconds, same constants (:7-17), same output file format (12 floats, 3x4 row-major,
// space separated, default ostream precision, :171-180).  Host code is C++ on the façade
// include/stocs.hpp; all hot-path work runs in libstocs_hip.so on the GPU.
//
// Inputs are flat cloud files (.stcl, written by model_matching_amd/cloudio.py) instead of
// rgb/depth/probability PNGs + model_search.ply + ppf_map: scene ingest and model preprocessing are
// outside the hot path.
//
// usage: stocs_single <scene.stcl> <model.stcl> [--edge edge.u8] [--seed N] [--out pose.txt]
//                     [--bases 100] [--max-sets 200] [--device 0] [--dbg DIR] [--cluster 1]
// --dbg DIR writes best_pose.ply / scene.ply as stocs_estimator::visualize_best_pose does (reference
// include/stocs.hpp:136-149); --cluster 1 additionally runs clustering::greedy_clustering (reference
// src/pose_clustering.cpp:79-121, which has no caller in the reference) on the scored candidates.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "../../include/stocs.hpp"

using micro = std::chrono::microseconds;

// reference stocs_match_one_object.cpp:7-17
static float distance_threshold = 0.005f;
static int ppf_tr_discretization = 5;
static int ppf_rot_discretization = 5;
static float edge_threshold = 0;
static float class_threshold = 0.10f;
static float sample_dispersion = 0.9f;
static int number_of_bases = 100;
static int maximum_congruent_sets = 200;
static int image_width = 640, image_height = 480;

// ASCII PLY with positions and normals (stand-in for rgbd::save_as_ply, reference src/rgbd.cpp:35-56)
static bool write_ply(const std::string& path, const std::vector<float>& pos, const std::vector<float>& nrm) {
    std::ofstream f(path);
    if (!f) return false;
    const size_t n = pos.size() / 3;
    f << "ply\nformat ascii 1.0\nelement vertex " << n << "\nproperty float x\nproperty float y\nproperty float z\n"
      << "property float normal_x\nproperty float normal_y\nproperty float normal_z\nend_header\n";
    for (size_t i = 0; i < n; ++i)
        f << pos[3 * i] << " " << pos[3 * i + 1] << " " << pos[3 * i + 2] << " " << nrm[3 * i] << " " << nrm[3 * i + 1] << " " << nrm[3 * i + 2] << "\n";
    return (bool)f;
}

static bool read_stcl(const std::string& path, std::vector<float>& pos, std::vector<float>& nrm, std::vector<float>* prob,
                      std::vector<int32_t>* pixel) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[8];
    int32_t n = 0, flags = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && std::string(magic, 8) == "STOCSCL1" && fread(&n, 4, 1, f) == 1 && fread(&flags, 4, 1, f) == 1 && n >= 0;
    if (ok) {
        pos.resize((size_t)n * 3); nrm.resize((size_t)n * 3);
        ok = fread(pos.data(), 4, pos.size(), f) == pos.size() && fread(nrm.data(), 4, nrm.size(), f) == nrm.size();
        if (ok && (flags & 1)) { std::vector<float> p(n); ok = fread(p.data(), 4, n, f) == (size_t)n; if (prob) *prob = p; }
        else if (prob) prob->assign(n, 1.0f);
        if (ok && (flags & 2)) { std::vector<int32_t> px((size_t)n * 2); ok = fread(px.data(), 4, px.size(), f) == px.size(); if (pixel) *pixel = px; }
    }
    fclose(f);
    return ok;
}

struct BaseGraph {  // reference stocs_match_one_object.cpp:26-48
    std::vector<int> baseIds_;
    float invariant1_, invariant2_;
    std::vector<stocs::Quadrilateral> congruent_quads;
    BaseGraph(const std::vector<int>& ids, float i1, float i2) : baseIds_(ids.begin(), ids.begin() + 4), invariant1_(i1), invariant2_(i2) {}
};

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cout << "usage: stocs_single <scene.stcl> <model.stcl> [--edge edge.u8] [--seed N] [--out pose.txt]" << std::endl;
        return -1;
    }
    std::string edge_path, out_path = "best_pose_candidate.txt", dbg_dir;
    int do_cluster = 0;
    uint64_t seed = 1;
    int device = -1;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--edge") edge_path = argv[i + 1];
        else if (k == "--seed") seed = strtoull(argv[i + 1], NULL, 10);
        else if (k == "--out") out_path = argv[i + 1];
        else if (k == "--bases") number_of_bases = atoi(argv[i + 1]);
        else if (k == "--max-sets") maximum_congruent_sets = atoi(argv[i + 1]);
        else if (k == "--device") device = atoi(argv[i + 1]);
        else if (k == "--dbg") dbg_dir = argv[i + 1];
        else if (k == "--cluster") do_cluster = atoi(argv[i + 1]);
    }
    stocs::SceneCloud scene;
    stocs::ModelCloud model;
    if (!read_stcl(argv[1], scene.pos, scene.nrm, &scene.class_probability, &scene.pixel)) { std::cerr << "cannot read scene " << argv[1] << std::endl; return 1; }
    if (!read_stcl(argv[2], model.pos, model.nrm, NULL, NULL)) { std::cerr << "cannot read model " << argv[2] << std::endl; return 1; }
    if (!edge_path.empty()) {
        std::ifstream ef(edge_path, std::ios::binary);
        scene.edge_map.resize((size_t)image_width * image_height);
        if (!ef.read((char*)scene.edge_map.data(), (std::streamsize)scene.edge_map.size())) { std::cerr << "cannot read edge map" << std::endl; return 1; }
    }
    std::cout << "|M| = " << model.size() << std::endl;
    std::cout << "|S|: " << scene.size() << std::endl;

    std::unique_ptr<stocs::stocs_estimator> est;
    try {
        est.reset(new stocs::stocs_estimator(model, scene, "dbg", image_width, image_height, distance_threshold, ppf_tr_discretization,
                                             ppf_rot_discretization, edge_threshold, class_threshold, device));
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;  // no GPU => loud failure, never a CPU fallback
        return 2;
    }
    stocs::stocs_estimator& stocs_ptr = *est;
    stocs_ptr.set_seed(seed);
    std::vector<BaseGraph> base_set;

    // Step 1: sample n bases on the scene (:79-105)
    auto start = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < number_of_bases; i++) {
        std::vector<int> base_indices(4, -1);
        float invariant1 = 0, invariant2 = 0;
        bool valid_base_found;
        if (stocs_ptr.has_edge_map()) valid_base_found = stocs_ptr.sample_instance_base(base_indices, invariant1, invariant2, sample_dispersion, i + 1);
        else valid_base_found = stocs_ptr.sample_class_base(base_indices, invariant1, invariant2);
        if (valid_base_found) base_set.emplace_back(base_indices, invariant1, invariant2);
    }
    auto finish = std::chrono::high_resolution_clock::now();
    std::cout << "Sampled " << base_set.size() << " bases in " << std::chrono::duration_cast<micro>(finish - start).count() << " microseconds\n";

    // Step 2 + 3: congruent sets and rigid transforms (:107-151)
    start = std::chrono::high_resolution_clock::now();
    for (auto& b : base_set) stocs_ptr.find_congruent_sets_on_model(b.baseIds_, b.invariant1_, b.invariant2_, &b.congruent_quads);
    int total_congruent_set_found = 0, base_number = 0;
    for (auto& b : base_set) {
        const int congruent_set_size = (int)b.congruent_quads.size();
        if (congruent_set_size < maximum_congruent_sets) {
            for (int i = 0; i < congruent_set_size; i++) stocs_ptr.get_rigid_transform_from_congruent_pair(b.baseIds_, b.congruent_quads[i], base_number);
        } else {
            // seeded sample without replacement (divergence Q5 from the 2N-vector random_shuffle, :134-142)
            std::vector<int> perm(congruent_set_size);
            for (int i = 0; i < congruent_set_size; i++) perm[i] = i;
            uint64_t z = seed * 0x9E3779B97F4A7C15ull + (uint64_t)base_number;
            for (int j = 0; j < maximum_congruent_sets; ++j) {
                z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
                const int k = j + (int)(z % (uint64_t)(congruent_set_size - j));
                std::swap(perm[j], perm[k]);
                stocs_ptr.get_rigid_transform_from_congruent_pair(b.baseIds_, b.congruent_quads[perm[j]], base_number);
            }
        }
        total_congruent_set_found += congruent_set_size;
        base_number++;
    }
    finish = std::chrono::high_resolution_clock::now();
    std::cout << "found " << total_congruent_set_found << " congruent sets in " << std::chrono::duration_cast<micro>(finish - start).count() << " microseconds\n";

    // Step 4: verify (:155-163)
    start = std::chrono::high_resolution_clock::now();
    std::cout << "Transforms to verify: " << stocs_ptr.get_all_transforms().size() << std::endl;
    stocs_ptr.compute_best_transform();
    finish = std::chrono::high_resolution_clock::now();
    std::cout << "evaluated transforms in " << std::chrono::duration_cast<micro>(finish - start).count() << " microseconds\n";
    std::cout << "maximum score: " << stocs_ptr.get_best_score() << std::endl;

    stocs::PoseCandidate* best_pose = stocs_ptr.get_best_pose();
    if (best_pose != NULL) {  // :171-180
        std::ofstream out_file_ptr(out_path, std::ofstream::out);
        const stocs::Mat4f& t = best_pose->transform;
        out_file_ptr << t(0, 0) << " " << t(0, 1) << " " << t(0, 2) << " " << t(0, 3) << " " << t(1, 0) << " " << t(1, 1) << " " << t(1, 2) << " "
                     << t(1, 3) << " " << t(2, 0) << " " << t(2, 1) << " " << t(2, 2) << " " << t(2, 3) << std::endl;
        if (!dbg_dir.empty()) {  // visualize_best_pose (stocs.hpp:136-149): model under the best camera-frame pose + the scene
            std::vector<float> mp(model.pos.size()), mn(model.nrm.size());
            for (size_t i = 0; i < model.pos.size() / 3; ++i)
                for (int r = 0; r < 3; ++r) {
                    mp[3 * i + r] = t(r, 0) * model.pos[3 * i] + t(r, 1) * model.pos[3 * i + 1] + t(r, 2) * model.pos[3 * i + 2] + t(r, 3);
                    mn[3 * i + r] = t(r, 0) * model.nrm[3 * i] + t(r, 1) * model.nrm[3 * i + 1] + t(r, 2) * model.nrm[3 * i + 2];
                }
            write_ply(dbg_dir + "/best_pose.ply", mp, mn);
            write_ply(dbg_dir + "/scene.ply", scene.pos, scene.nrm);
        }
        if (do_cluster) {  // greedy_clustering(hypotheses, 0.8, best, 10, 2 cm, 15 deg, no symmetry)
            std::vector<stocs::PoseCandidate*> all = stocs_ptr.get_pose_candidates();
            std::vector<float> poses(all.size() * 16), lcp(all.size());
            for (size_t i = 0; i < all.size(); ++i) { std::memcpy(&poses[16 * i], all[i]->transform.data(), 64); lcp[i] = all[i]->lcp; }
            std::vector<int32_t> keep(all.size() + 1);
            const float sym[3] = {0, 0, 0};
            int nk = 0;
            stocs_cluster_poses(poses.data(), lcp.data(), (int)all.size(), 0.8f, stocs_ptr.get_best_score(), 10, 0.02f, 15.0f, sym, keep.data(), (int)keep.size(), &nk);
            std::cout << "clustered hypotheses: " << nk << std::endl;
            for (int i = 0; i < nk; ++i) std::cout << "  cluster " << i << ": candidate " << keep[i] << " lcp " << lcp[keep[i]] << std::endl;
        }
    } else {
        std::cout << "no pose found" << std::endl;
    }
    return 0;
}
