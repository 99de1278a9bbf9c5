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

// How a tree is built.  The surface-area cost model counts one visit of a
// four-wide node as 1 and one triangle test as `tri_cost`.
struct BvhTuning {
    int bins = 32;              // centroid bins per axis for large ranges
    uint32_t sweep_below = 64;  // ranges up to this size try every split of the sorted centroids instead
    float tri_cost = 0.3f;      // measured on the MI355X: 0.3 ... 0.7 render within 1.5 % of each other, 0.3 fastest
    uint32_t max_leaf = 4;      // most triangles a leaf may hold (<= 8); what the library passes to build_bvh
    bool collapse = true;       // leaves and wide-node children chosen by minimum surface-area cost;
                                // false: every range of <= max_leaf triangles is a leaf, widest-box-first widening
    uint32_t rotation_rounds = 0; // rounds of tree rotations over the binary tree before leaves are chosen (find nothing on uniform meshes)
};
// Defaults, each overridable by an environment variable (PATHTRACING_AMD_BVH_BINS, _SWEEP, _TRI_COST,
// _MAX_LEAF, _COLLAPSE, _ROTATE) so that measuring scripts can compare trees without rebuilding the library.
BvhTuning default_bvh_tuning();

// Surface-area-heuristic BVH2, at most `max_leaf` (<= 8) triangles per leaf, depth
// bounded by ptk::kBvhMaxDepth.
BuiltBvh build_bvh(const float* positions, uint64_t triangle_count, uint32_t max_leaf = 4);
BuiltBvh build_bvh(const float* positions, uint64_t triangle_count, uint32_t max_leaf, const BvhTuning& tuning);

// Conservative bounds of one triangle: its AABB grown by the distance within
// which the reference's f32 Moller-Trumbore test may still accept a ray that
// geometrically misses the triangle.
void padded_triangle_bounds(const float* p9, float reach, float* bmin, float* bmax);
// Largest distance between a ray origin and the mesh that the padding budgets for.
float padding_reach(const float* mesh_bmin, const float* mesh_bmax);

// The device form of a tree (ptk::WideNode): every node takes descendants of the
// binary node it stands for as its (at most four) children -- the set that makes
// the expected cost of a search smallest -- and child boxes are quantised outwards.
struct WideBvh {
    std::vector<ptk::WideNode> nodes;
    int32_t root_ref = ptk::kBvhSentinel; // node 0, or the leaf of a tiny mesh
    uint32_t max_stack = 0;               // most entries a depth-first search can have pending
};
WideBvh widen_bvh(const BuiltBvh& bvh);
WideBvh widen_bvh(const BuiltBvh& bvh, const BvhTuning& tuning);

std::vector<uint8_t> serialize_tree(const BuiltBvh& bvh, const float* positions, uint64_t triangle_count);
// Returns false (and leaves `out` untouched) when the blob is empty, malformed,
// built for other geometry, or not conservative for this mesh.
bool deserialize_tree(const uint8_t* data, uint64_t size, const float* positions, uint64_t triangle_count,
                      BuiltBvh* out);

} // namespace pt
