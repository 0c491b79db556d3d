// pose_clustering.hpp -- the public surface of the reference's include/pose_clustering.hpp:9-28 on the C ABI of
// libstocs_hip.so (stocs_cluster_poses, stocs_icp_point_to_plane).  PCL cloud pointers become std::vector<Point3D>,
// Eigen types the stocs:: ones of stocs.hpp; trimmed_icp is declared but never defined in the reference and is not offered.
#ifndef STOCS_POSE_CLUSTERING_HPP
#define STOCS_POSE_CLUSTERING_HPP

#include "stocs.hpp"

namespace clustering {

// reference src/pose_clustering.cpp:79-121
inline void greedy_clustering(std::vector<PoseCandidate*>& hypotheses_set, float acceptable_fraction, float best_score, int maximum_pose_count,
                              float min_distance, float min_angle, VectorType sym_info, std::vector<PoseCandidate*>& clustered_hypotheses_set) {
    const int n = (int)hypotheses_set.size();
    std::vector<float> poses((size_t)n * 16), lcp((size_t)n);
    for (int i = 0; i < n; ++i) { std::memcpy(&poses[(size_t)i * 16], hypotheses_set[(size_t)i]->transform.data(), 64); lcp[(size_t)i] = hypotheses_set[(size_t)i]->lcp; }
    std::vector<int32_t> keep((size_t)n + 1);
    int nk = 0;
    if (stocs_cluster_poses(poses.data(), lcp.data(), n, acceptable_fraction, best_score, maximum_pose_count, min_distance, min_angle, sym_info.data(), keep.data(),
                            (int)keep.size(), &nk) != STOCS_OK)
        return;
    for (int i = 0; i < nk; ++i) clustered_hypotheses_set.push_back(hypotheses_set[(size_t)keep[(size_t)i]]);
}

// reference src/pose_clustering.cpp:123-140 (PCL IterativeClosestPointWithNormals, 5 iterations, 3.5 cm): aligns `segment`
// onto `model` (which carries the normals); parity with PCL unpinned
inline void point_to_plane_icp(const std::vector<Point3D>& segment, const std::vector<Point3D>& model, MatrixType& offset_transform) {
    std::vector<float> s(segment.size() * 3), t(model.size() * 3), tn(model.size() * 3);
    for (size_t i = 0; i < segment.size(); ++i) for (int k = 0; k < 3; ++k) s[3 * i + k] = segment[i].pos()[k];
    for (size_t i = 0; i < model.size(); ++i) for (int k = 0; k < 3; ++k) { t[3 * i + k] = model[i].pos()[k]; tn[3 * i + k] = model[i].normal()[k]; }
    int nc = 0;
    (void)stocs_icp_point_to_plane(s.data(), (int)segment.size(), t.data(), tn.data(), (int)model.size(), 5, 0.035f, -1, offset_transform.data(), &nc);
}

}  // namespace clustering
#endif
