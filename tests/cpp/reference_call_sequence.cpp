// The call sequence of the reference's only caller -- run_stocs_estimation + main, reference
// src/stocs_match_one_object.cpp:51-215 -- RESTATED (not copied) against this repo's include/stocs.hpp and
// include/pose_clustering.hpp: the same constructor argument list, the same per-base method calls with the reference's
// types (Point3D, Quadrilateral, PoseCandidate, PPFMapType at global scope), the same output file.  That it compiles and
// links without touching the call sites is the drop-in claim of the façade; tests/test_abi_cpu.py builds it, and
// tests/test_driver_gpu.py runs it on the GPU box next to the batched driver.
//
// The only liberties: the object-map directory comes from argv / STOCS_REPO_PATH instead of a path compiled into the
// binary, camera constants can be overridden by environment variables (the reference edits them in the source), and the
// subset of a base with >= 200 congruent sets is drawn with a seeded std::shuffle (the reference's std::random_shuffle of
// a 2N-element vector is unseeded and biased, divergence Q5).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <random>

#include <pose_clustering.hpp>
#include <stocs.hpp>

namespace {

std::string repo_path = ".";
float voxel_size = 0.005f, distance_threshold = 0.005f, edge_threshold = 0, class_threshold = 0.10f, sample_dispersion = 0.9f;
int ppf_tr_discretization = 5, ppf_rot_discretization = 5, number_of_bases = 100, maximum_congruent_sets = 200;
std::vector<float> cam_intrinsics = {1066.778f, 312.986f, 1067.487f, 241.310f};
float depth_scale = 1 / 10000.0f;
int image_width = 640, image_height = 480;

struct BaseGraph {
    std::vector<int> baseIds_;
    float invariant1_, invariant2_;
    std::vector<Quadrilateral> congruent_quads;
    BaseGraph(const std::vector<int>& ids, float i1, float i2) : baseIds_(ids.begin(), ids.begin() + 4), invariant1_(i1), invariant2_(i2) {}
};

void run_stocs_estimation(const std::string& scene_path, const std::string& object_name, PPFMapType& ppf_map_preloaded) {
    const std::string edge_probability_path = scene_path + "/probability_maps/edge.png";
    stocs::stocs_estimator stocs_ptr(repo_path + "/models/" + object_name + "/model_search.ply", ppf_map_preloaded, scene_path + "/rgb.png",
                                     scene_path + "/depth.png", scene_path + "/probability_maps/" + object_name + ".png", edge_probability_path,
                                     scene_path + "/dbg", cam_intrinsics, image_width, image_height, depth_scale, 1.0f, voxel_size, distance_threshold,
                                     ppf_tr_discretization, ppf_rot_discretization, edge_threshold, class_threshold);
    if (const char* s = getenv("STOCS_SEED")) stocs_ptr.set_seed(strtoull(s, NULL, 10));

    // 1: bases, one call per attempt
    std::vector<BaseGraph> base_set;
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < number_of_bases; i++) {
        std::vector<int> base_indices(4, -1);
        float invariant1 = 0, invariant2 = 0;
        std::vector<Point3D> segment;
        bool found;
        if (stocs::file_exists(edge_probability_path)) found = stocs_ptr.sample_instance_base(base_indices, invariant1, invariant2, segment, sample_dispersion, i + 1);
        else found = stocs_ptr.sample_class_base(base_indices, invariant1, invariant2);
        if (found) base_set.emplace_back(base_indices, invariant1, invariant2);
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    std::cout << "Sampled " << base_set.size() << " bases in " << std::chrono::duration_cast<micro>(t1 - t0).count() << " microseconds\n";

    // 2: congruent sets per base;  3: at most maximum_congruent_sets transforms per base
    t0 = std::chrono::high_resolution_clock::now();
    for (auto& b : base_set) stocs_ptr.find_congruent_sets_on_model(b.baseIds_, b.invariant1_, b.invariant2_, &b.congruent_quads);
    long long total_congruent_set_found = 0;
    int base_number = 0;
    std::mt19937 shuffle_rng(12345);
    for (auto& b : base_set) {
        const int congruent_set_size = (int)b.congruent_quads.size();
        std::vector<int> order(congruent_set_size);
        for (int i = 0; i < congruent_set_size; i++) order[i] = i;
        if (congruent_set_size >= maximum_congruent_sets) std::shuffle(order.begin(), order.end(), shuffle_rng);
        for (int i = 0; i < std::min(congruent_set_size, maximum_congruent_sets); i++)
            stocs_ptr.get_rigid_transform_from_congruent_pair(b.baseIds_, b.congruent_quads[order[i]], base_number);
        total_congruent_set_found += congruent_set_size;
        base_number++;
    }
    t1 = std::chrono::high_resolution_clock::now();
    std::cout << "found " << total_congruent_set_found << " congruent sets in " << std::chrono::duration_cast<micro>(t1 - t0).count() << " microseconds\n";

    // 4: verification
    t0 = std::chrono::high_resolution_clock::now();
    stocs_ptr.compute_best_transform();
    t1 = std::chrono::high_resolution_clock::now();
    std::cout << "evaluated transforms in " << std::chrono::duration_cast<micro>(t1 - t0).count() << " microseconds\n";

    stocs_ptr.visualize_best_pose();
    PoseCandidate* best_pose = stocs_ptr.get_best_pose();
    if (best_pose != NULL) {
        std::ofstream out_file_ptr(scene_path + "/best_pose_candidate_" + object_name + ".txt", std::ofstream::out);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) out_file_ptr << best_pose->transform(r, c) << (r == 2 && c == 3 ? "" : " ");
        out_file_ptr << std::endl;
        // pose post-processing (library API without a caller in the reference): greedy clustering of the scored hypotheses
        std::vector<PoseCandidate*> all = stocs_ptr.get_pose_candidates(), clustered;
        clustering::greedy_clustering(all, 0.8f, stocs_ptr.get_best_score(), 10, 0.02f, 15.0f, VectorType(0, 0, 0), clustered);
        std::cout << "candidates " << all.size() << ", clustered " << clustered.size() << ", best lcp " << best_pose->lcp << std::endl;
    } else {
        std::cout << "no pose found" << std::endl;
    }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::cout << "Enter scene path and object name as arguments!" << std::endl;
        return -1;
    }
    if (const char* e = getenv("STOCS_REPO_PATH")) repo_path = e;
    if (const char* e = getenv("STOCS_INTRINSICS")) sscanf(e, "%f,%f,%f,%f", &cam_intrinsics[0], &cam_intrinsics[1], &cam_intrinsics[2], &cam_intrinsics[3]);
    if (const char* e = getenv("STOCS_DEPTH_SCALE")) depth_scale = (float)atof(e);
    const std::string scene_path = argv[1], object_name = argv[2];
    PPFMapType model_map;
    rgbd::load_ppf_map(repo_path + "/models/" + object_name + "/ppf_map", model_map);
    if (system(("rm -rf '" + scene_path + "/dbg' && mkdir '" + scene_path + "/dbg'").c_str()) != 0) return 1;
    try {
        run_stocs_estimation(scene_path, object_name, model_map);
    } catch (const std::exception& e) {   // construction failed (no GPU, missing file): loud, never a CPU fallback
        std::cerr << e.what() << std::endl;
        return 2;
    }
    return 0;
}
