// stocs_single -- the repo's equivalent of the reference driver (reference src/stocs_match_one_object.cpp:51-215): same
// command line (scene directory + object name), same input files (depth.png, probability_maps/<object>.png, optional
// probability_maps/edge.png, models/<object>/model_search.ply + ppf_map), same four phases, same three timed spans printed
// in microseconds, same constants (:7-17), same output file (12 floats, 3x4 row-major, space separated, default ostream
// precision, :171-180), best_pose.ply / scene.ply in <scene>/dbg (stocs.hpp:136-149).  Host code is C++ on the façade
// include/stocs.hpp; all hot-path work runs in libstocs_hip.so on the GPU.
//
// The loops of the reference's caller (one call per base, :81-147) are taken through the façade's batched methods: all
// 100 attempts, all bases' congruent sets and all <= 200-per-base transforms are one GPU pass each, and the subset of a
// base with >= 200 congruent sets is the library's seeded rule (stocs_make_transforms).  The per-call spelling of the same
// sequence is tests/cpp/reference_call_sequence.cpp.
//
//   stocs_single <scene_path> <object_name> [--repo DIR] [--intrinsics fx,cx,fy,cy] [--depth-scale S] [--voxel V] ...
//   stocs_single --clouds <scene.stcl> <model.stcl> [--edge edge.u8] ...       (flat clouds, e.g. the synthetic workloads)
// common options: --seed N --bases 100 --max-sets 200 --out FILE --dbg DIR --cluster 1
// The reference edits its per-data-set constants in the source (README.md:42-66); here they are options with the
// reference's YCB values as defaults.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "../../include/pose_clustering.hpp"
#include "../../include/stocs.hpp"

// reference stocs_match_one_object.cpp:4-24
static std::string repo_path = ".";
static float voxel_size = 0.005f;
static float distance_threshold = 0.005f;
static int ppf_tr_discretization = 5;
static int ppf_rot_discretization = 5;
static float edge_threshold = 0;
static float class_threshold = 0.10f;
static float sample_dispersion = 0.9f;
static int number_of_bases = 100;
static int maximum_congruent_sets = 200;
static std::vector<float> cam_intrinsics = {1066.778f, 312.986f, 1067.487f, 241.310f};  // YCB
static float depth_scale = 1 / 10000.0f;
static int image_width = 640, image_height = 480;

static bool read_stcl(const std::string& path, std::vector<float>& pos, std::vector<float>& nrm, std::vector<float>* prob, std::vector<int32_t>* pixel) {
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

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cout << "Enter scene path and object name as arguments!" << std::endl;   // :190
        return -1;
    }
    const bool clouds = std::string(argv[1]) == "--clouds";
    if (clouds && argc < 4) { std::cout << "usage: stocs_single --clouds <scene.stcl> <model.stcl> [options]" << std::endl; return -1; }
    const std::string a1 = argv[clouds ? 2 : 1], a2 = argv[clouds ? 3 : 2];
    if (const char* e = getenv("STOCS_REPO_PATH")) repo_path = e;
    std::string edge_path, out_path, dbg_dir;
    int do_cluster = 0, n_trials = 0;
    uint64_t seed = 1;
    for (int i = clouds ? 4 : 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--edge") edge_path = v;
        else if (k == "--seed") seed = strtoull(v.c_str(), NULL, 10);
        else if (k == "--out") out_path = v;
        else if (k == "--bases") number_of_bases = atoi(v.c_str());
        else if (k == "--max-sets") maximum_congruent_sets = atoi(v.c_str());
        else if (k == "--dbg") dbg_dir = v;
        else if (k == "--cluster") do_cluster = atoi(v.c_str());
        else if (k == "--trials") n_trials = atoi(v.c_str());   // N independent runs (seeds seed, seed + 1, ...) in one set of GPU launches; the best one is written
        else if (k == "--repo") repo_path = v;
        else if (k == "--voxel") voxel_size = (float)atof(v.c_str());
        else if (k == "--depth-scale") depth_scale = (float)atof(v.c_str());
        else if (k == "--class-threshold") class_threshold = (float)atof(v.c_str());
        else if (k == "--intrinsics") {
            if (sscanf(v.c_str(), "%f,%f,%f,%f", &cam_intrinsics[0], &cam_intrinsics[1], &cam_intrinsics[2], &cam_intrinsics[3]) != 4) { std::cerr << "--intrinsics fx,cx,fy,cy" << std::endl; return -1; }
        } else { std::cerr << "unknown option " << k << std::endl; return -1; }
    }

    std::unique_ptr<stocs::stocs_estimator> est;
    try {
        if (clouds) {
            stocs::SceneCloud scene;
            stocs::ModelCloud model;
            if (!read_stcl(a1, scene.pos, scene.nrm, &scene.class_probability, &scene.pixel)) { std::cerr << "cannot read scene " << a1 << std::endl; return 1; }
            if (!read_stcl(a2, model.pos, model.nrm, NULL, NULL)) { std::cerr << "cannot read model " << a2 << std::endl; return 1; }
            if (!edge_path.empty()) {
                std::ifstream ef(edge_path, std::ios::binary);
                scene.edge_map.resize((size_t)image_width * image_height);
                if (!ef.read((char*)scene.edge_map.data(), (std::streamsize)scene.edge_map.size())) { std::cerr << "cannot read edge map" << std::endl; return 1; }
            }
            if (out_path.empty()) out_path = "best_pose_candidate.txt";
            std::cout << "|M| = " << model.size() << std::endl;
            est.reset(new stocs::stocs_estimator(model, scene, dbg_dir, image_width, image_height, distance_threshold, ppf_tr_discretization, ppf_rot_discretization,
                                                 edge_threshold, class_threshold));
        } else {
            // :56-62, :196-208
            const std::string scene_path = a1, object_name = a2;
            const std::string rgb_path = scene_path + "/rgb.png", depth_path = scene_path + "/depth.png";
            const std::string class_probability_path = scene_path + "/probability_maps/" + object_name + ".png";
            const std::string edge_probability_path = scene_path + "/probability_maps/edge.png";
            const std::string model_path = repo_path + "/models/" + object_name + "/model_search.ply";
            if (out_path.empty()) out_path = scene_path + "/best_pose_candidate_" + object_name + ".txt";
            const bool own_dbg = dbg_dir.empty();
            if (own_dbg) dbg_dir = scene_path + "/dbg";
            std::cout << "############# LOADING OBJECT MAPS ################" << std::endl;
            PPFMapType model_map;
            rgbd::load_ppf_map(repo_path + "/models/" + object_name + "/ppf_map", model_map);
            std::cout << "############# LOADING OBJECT COMPLETE ################" << std::endl;
            // :207-208 (only the reference's own <scene>/dbg is wiped; a directory named with --dbg is merely created)
            if (system(((own_dbg ? "rm -rf '" + dbg_dir + "' && " : std::string()) + "mkdir -p '" + dbg_dir + "'").c_str()) != 0) std::cerr << "cannot create " << dbg_dir << std::endl;
            std::cout << "############# RUNNING STOCS for Scene: " << scene_path << ", Object: " << object_name << " ##############" << std::endl;
            est.reset(new stocs::stocs_estimator(model_path, model_map, rgb_path, depth_path, class_probability_path, edge_probability_path, dbg_dir, cam_intrinsics,
                                                 image_width, image_height, depth_scale, 1.0f, voxel_size, distance_threshold, ppf_tr_discretization,
                                                 ppf_rot_discretization, edge_threshold, class_threshold));
        }
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;  // no GPU => loud failure, never a CPU fallback
        return 2;
    }
    stocs::stocs_estimator& stocs_ptr = *est;
    stocs_ptr.set_seed(seed);

    if (n_trials > 0) {
        // BASELINE config 4: N independent StoCS trials -- each the whole loop of run_stocs_estimation (:79-165) with its own seed --
        // in ONE set of launches (stocs_run_trials); the best pose over the trials is the result
        std::vector<stocs::stocs_estimator::TrialResult> res;
        auto t0 = std::chrono::high_resolution_clock::now();
        const int best = stocs_ptr.run_trials(n_trials, seed, number_of_bases, maximum_congruent_sets, sample_dispersion, &res);
        auto t1 = std::chrono::high_resolution_clock::now();
        long long cand = 0;
        for (size_t t = 0; t < res.size(); ++t) {
            cand += res[t].n_candidates;
            std::cout << "trial " << t << ": bases " << res[t].n_bases << " congruent sets " << res[t].n_congruent_sets << " candidates " << res[t].n_candidates
                      << " best lcp " << res[t].best_lcp << std::endl;
        }
        const long long us = (long long)std::chrono::duration_cast<micro>(t1 - t0).count();
        char tl[256];
        snprintf(tl, sizeof(tl), "trials: n=%d best_trial=%d best_lcp=%.9g candidates=%lld total_microseconds=%lld", n_trials, best,
                 best >= 0 ? (double)res[(size_t)best].best_lcp : 0.0, cand, us);
        if (PoseCandidate* bp = stocs_ptr.get_best_trial_pose()) {
            std::ofstream o(out_path, std::ofstream::out);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) o << bp->transform(r, c) << (r == 2 && c == 3 ? "" : " ");
            o << std::endl;
            std::cout << "pose:";
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { char b[32]; snprintf(b, sizeof(b), " %.9g", (double)bp->transform(r, c)); std::cout << b; }
            std::cout << std::endl;
        } else {
            std::cout << "no pose found" << std::endl;
        }
        std::cout << tl << std::endl;
        return 0;
    }

    // Step 1: sample n bases on the scene (:79-105)
    auto start = std::chrono::high_resolution_clock::now();
    const int n_bases = stocs_ptr.sample_bases(number_of_bases, sample_dispersion);
    auto finish = std::chrono::high_resolution_clock::now();
    std::cout << "Sampled " << n_bases << " bases in " << std::chrono::duration_cast<micro>(finish - start).count() << " microseconds\n";
    auto total_time = std::chrono::duration_cast<micro>(finish - start).count();

    // Step 2 + 3: congruent sets and rigid transforms (:107-153)
    start = std::chrono::high_resolution_clock::now();
    const long long total_congruent_set_found = stocs_ptr.find_congruent_sets_all();
    const int n_candidates = stocs_ptr.make_transforms(maximum_congruent_sets);
    finish = std::chrono::high_resolution_clock::now();
    std::cout << "found " << total_congruent_set_found << " congruent sets in " << std::chrono::duration_cast<micro>(finish - start).count() << " microseconds\n";
    total_time += std::chrono::duration_cast<micro>(finish - start).count();

    // Step 4: verify all transforms to get the best pose (:155-165)
    start = std::chrono::high_resolution_clock::now();
    stocs_ptr.compute_best_transform();
    finish = std::chrono::high_resolution_clock::now();
    std::cout << "evaluated transforms in " << std::chrono::duration_cast<micro>(finish - start).count() << " microseconds\n";
    total_time += std::chrono::duration_cast<micro>(finish - start).count();

    PoseCandidate* best_pose = stocs_ptr.get_best_pose();
    char line[256];
    snprintf(line, sizeof(line), "summary: bases=%d congruent_sets=%lld candidates=%d best_lcp=%.9g best_index=%d total_microseconds=%lld", n_bases,
             total_congruent_set_found, n_candidates, (double)stocs_ptr.get_best_score(), stocs_ptr.get_best_index(), (long long)total_time);
    if (!dbg_dir.empty()) stocs_ptr.visualize_best_pose();   // :167
    if (best_pose != NULL) {  // :171-180
        std::ofstream out_file_ptr;
        out_file_ptr.open(out_path, std::ofstream::out);
        out_file_ptr << best_pose->transform(0, 0) << " " << best_pose->transform(0, 1) << " " << best_pose->transform(0, 2) << " " << best_pose->transform(0, 3) << " "
                     << best_pose->transform(1, 0) << " " << best_pose->transform(1, 1) << " " << best_pose->transform(1, 2) << " " << best_pose->transform(1, 3) << " "
                     << best_pose->transform(2, 0) << " " << best_pose->transform(2, 1) << " " << best_pose->transform(2, 2) << " " << best_pose->transform(2, 3) << std::endl;
        out_file_ptr.close();
        // full-precision copy of the same 12 numbers for tools that compare poses
        std::cout << "pose:";
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { char b[32]; snprintf(b, sizeof(b), " %.9g", (double)best_pose->transform(r, c)); std::cout << b; }
        std::cout << std::endl;
        if (do_cluster) {  // clustering::greedy_clustering (pose_clustering.cpp:79-121; no caller in the reference): 0.8, best, 10, 2 cm, 15 deg, no symmetry
            std::vector<PoseCandidate*> all = stocs_ptr.get_pose_candidates(), kept;
            clustering::greedy_clustering(all, 0.8f, stocs_ptr.get_best_score(), 10, 0.02f, 15.0f, VectorType(0, 0, 0), kept);
            std::cout << "clustered hypotheses: " << kept.size() << std::endl;
            for (size_t i = 0; i < kept.size(); ++i) std::cout << "  cluster " << i << ": base " << kept[i]->base_index << " lcp " << kept[i]->lcp << std::endl;
        }
    } else {
        std::cout << "no pose found" << std::endl;
    }
    std::cout << line << std::endl;
    return 0;
}
