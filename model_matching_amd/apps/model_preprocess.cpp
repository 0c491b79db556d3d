// model_preprocess -- the repo's equivalent of the reference's offline tool (reference src/model_preprocess.cpp:14-39):
// models/<object>/textured_vertices.ply -> models/<object>/model_search.ply + models/<object>/ppf_map, through
// stocs::pre_process_model (normals, voxel grid and the O(|M|^2) PPF index all on the GPU).
//   model_preprocess <object_name> [--repo DIR] [--voxel 0.01] [--normal-radius 0.005] [--model-scale 1.0]
// The reference edits these constants in the source per data set (README.md:42-60); the defaults are its YCB values.
#include <cstdlib>
#include <iostream>
#include <string>

#include "../../include/stocs.hpp"

static std::string repo_path = ".";
static float voxel_size = 0.01f;       // :7
static float normal_radius = 0.005f;   // :8
static float model_scale = 1.0f;       // :9
static int ppf_tr_discretization = 5;  // :12
static int ppf_rot_discretization = 5; // :13

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cout << "Enter name of the object model!!" << std::endl;   // :18
        return -1;
    }
    const std::string object_name = argv[1];
    if (const char* e = getenv("STOCS_REPO_PATH")) repo_path = e;
    for (int i = 2; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--repo") repo_path = v;
        else if (k == "--voxel") voxel_size = (float)atof(v.c_str());
        else if (k == "--normal-radius") normal_radius = (float)atof(v.c_str());
        else if (k == "--model-scale") model_scale = (float)atof(v.c_str());
        else { std::cerr << "unknown option " << k << std::endl; return -1; }
    }
    const std::string model_path = repo_path + "/models/" + object_name;
    remove((model_path + "/model_search.ply").c_str());   // :25-26
    remove((model_path + "/ppf_map").c_str());
    try {
        stocs::pre_process_model(model_path + "/textured_vertices.ply", normal_radius, model_scale, 1.0f, voxel_size, ppf_tr_discretization, ppf_rot_discretization,
                                 model_path + "/model_search.ply", model_path + "/ppf_map");
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;   // no GPU => loud failure
        return 2;
    }
    return 0;
}
