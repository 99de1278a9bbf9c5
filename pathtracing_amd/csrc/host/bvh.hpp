// Acceleration structure of one mesh: host-side build, validation and the
// ".tree" serialisation.  The reference has no counterpart: its kd_tree node is
// a stub (worker/src/scene/hierarchy/kd_tree.rs:47-51, scene/mod.rs:94-96) and
// scene_data/cube.tree is an empty file.  The contract the structure has to
// honour is the linear scan it replaces (mesh/mod.rs:53-65): same closest
// triangle, lowest index among equal t.
#pragma once
#include <cstdint>
#include <vector>

#include "../kernels/pt_scene_device.h"
#include "host_scene.hpp"

namespace pt {

struct BuiltBvh {
    std::vector<ptk::BvhNode> nodes; // empty when root_ref is a leaf
    std::vector<uint32_t> order;     // leaf order -> original triangle index
    int32_t root_ref = ptk::kBvhSentinel;
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    uint32_t max_depth = 0;
};

// Binned-SAH BVH2, at most `max_leaf` (<= 8) triangles per leaf, depth bounded
// by ptk::kBvhMaxDepth.
BuiltBvh build_bvh(const float* positions, uint64_t triangle_count, uint32_t max_leaf = 4);

// Conservative bounds of one triangle: its AABB grown by the distance within
// which the reference's f32 Moller-Trumbore test may still accept a ray that
// geometrically misses the triangle.
void padded_triangle_bounds(const float* p9, float reach, float* bmin, float* bmax);
// Largest distance between a ray origin and the mesh that the padding budgets for.
float padding_reach(const float* mesh_bmin, const float* mesh_bmax);

// The device form of a tree (ptk::WideNode): every node takes its grandchildren
// as children where that gives at most four, child boxes are quantised outwards.
struct WideBvh {
    std::vector<ptk::WideNode> nodes;
    int32_t root_ref = ptk::kBvhSentinel; // node 0, or the leaf of a tiny mesh
    uint32_t max_stack = 0;               // most entries a depth-first search can have pending
};
WideBvh widen_bvh(const BuiltBvh& bvh);

std::vector<uint8_t> serialize_tree(const BuiltBvh& bvh, const float* positions, uint64_t triangle_count);
// Returns false (and leaves `out` untouched) when the blob is empty, malformed,
// built for other geometry, or not conservative for this mesh.
bool deserialize_tree(const uint8_t* data, uint64_t size, const float* positions, uint64_t triangle_count,
                      BuiltBvh* out);

} // namespace pt
