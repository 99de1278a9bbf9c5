// Device-resident scene layout (plain structs, uploaded once per scene).
#pragma once
#include <cstddef>

#include "pt_vec.h"

namespace ptk {

// ---- acceleration structure of one mesh --------------------------------------
// 64-byte node = one cache-line-half fetch brings both children's boxes.  A child
// reference >= 0 is another node; < 0 is a leaf: ~ref = (first_triangle << 3) |
// (count - 1), count in 1..8, triangles addressed in leaf order.
struct alignas(64) BvhNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

constexpr int32_t kBvhSentinel = 0x7FFFFFFF; // "stack empty" / "no child"

// What the GPU searches: the same tree with every other level removed (up to four
// children per node) and the children's boxes quantised to 8 bits per coordinate
// on a grid local to the node, so that a node is one 64-byte fetch and a search
// makes half as many dependent fetches.  Child c's box is, per axis a,
//   [origin[a] + lo(a, c) * step[a], origin[a] + hi(a, c) * step[a]]
// with lo rounded down and hi rounded up (the box only grows); step[a] is a power
// of two.  Children are referenced like BvhNode's; unused slots hold kBvhSentinel.
// Nodes are stored in pre-order, so an inner child's index is always larger than
// its parent's: following child references cannot loop.
// Read as four 16-byte rows: (origin, step x) (qlo[3], qhi x) (qhi y, qhi z, step y, step z) (child[4]).
struct alignas(64) WideNode {
    float origin[3];
    float step_x;
    uint32_t qlo[3];    // byte c of qlo[a] = lo(a, c)
    uint32_t qhi[3];
    float step_yz[2];
    int32_t child[4];
};
static_assert(sizeof(WideNode) == 64 && offsetof(WideNode, child) == 48 && offsetof(WideNode, qlo) == 16 &&
                  offsetof(WideNode, qhi) == 28 && offsetof(WideNode, step_yz) == 40,
              "WideNode rows are read as four float4");
constexpr int kLeafMaxTris = 8;
constexpr int kBvhMaxDepth = 60; // builder guarantee; traversal stack holds 64

PT_HD int32_t bvh_make_leaf(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

// Triangle as the intersection loop reads it: three 16-byte rows
//   (v0.xyz, bits of original index) (v1.xyz, 0) (v2.xyz, 0)
struct alignas(16) LeafTri {
    float v0[3];
    uint32_t orig_index;
    float v1[3];
    uint32_t pad1;
    float v2[3];
    uint32_t pad2;
};
static_assert(sizeof(LeafTri) == 48, "LeafTri must be 48 bytes");

// Everything Triangle::trace_ray needs after the hit is known
// (triangle/mod.rs:90-127), indexed by the ORIGINAL triangle index.
struct alignas(16) ShadeTri {
    float true_normal[3];
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad[2];
};
static_assert(sizeof(ShadeTri) == 80, "ShadeTri must be 80 bytes");

struct DMesh {
    const WideNode* nodes;  // node 0 is the root (absent when root_ref is a leaf)
    const LeafTri* tris;    // leaf order
    const LeafTri* tris_orig; // original order (brute-force scan)
    const ShadeTri* shade;  // original order
    uint32_t tri_count;
    uint32_t node_count;
    int32_t root_ref;       // node 0, or a leaf reference for tiny meshes
    uint32_t pad;
    float bmin[3], bmax[3]; // padded bounds of the whole mesh (object space)
};

// ---- scene graph ------------------------------------------------------------
// Pre-order array; `end` = index one past the node's subtree.
struct DNode {
    uint32_t type;
    uint32_t end;
    uint32_t material_id;
    uint32_t mesh_id;
    uint32_t parent;  // nearest enclosing collection / transform node, or 0xFFFFFFFF
    uint32_t depth;   // number of enclosing collection / transform nodes
    uint32_t mesh_ordinal; // MESH: how many mesh nodes outside the instance run precede this one in the array
    uint32_t only_mesh;    // TRANSFORM whose subtree is one mesh node (behind kd_tree nodes or not): that
                           // mesh node's ordinal; 0xFFFFFFFF otherwise
    // TRANSFORM (transform.rs:33-40 plus the per-ray normal_matrix() of ray.rs:29)
    M4 matrix, matrix_inverse;
    M3 normal_matrix, inverse_normal_matrix;
    // SPHERE (sphere.rs:48-53)
    V3 center;
    float radius, radius_sqr;
    // PLANE (plane.rs:60-66), axes normalised
    V3 point, normal, tangent, bitangent;
};

struct DMaterial {
    uint32_t type;
    uint32_t color_is_texture;
    uint32_t uv_mode;
    uint32_t tex_w, tex_h;
    const uint8_t* texels; // RGBA8, top row first
    V3 color, emission;
    float refraction, reflective, emissive, refractive;
    V3 albedo;
    float roughness, metallic;
    V3 f0;               // pbr.rs:93
    float roughness_sqr; // pbr.rs:94
};

constexpr int kMaxGraphDepth = 8; // nested collection/transform frames the kernels keep

// One mesh node as a target of BVH searches: the transform nodes above it,
// outermost first, take a world-space ray into the mesh's space.
struct MeshTarget {
    uint32_t node_index;
    uint32_t mesh_id;
    uint32_t chain_len;
    uint32_t chain[kMaxGraphDepth];
};

// ---- instance run -----------------------------------------------------------
// A scene that places one mesh many times does it with a collection whose
// children are transform -> (kd_tree ->) mesh subtrees.  The reference visits
// them one after the other (node_collection.rs:43-55); here a run of such
// children (and of meshes placed without a transform) directly under the root collection gets a bounding-volume tree over
// the children's world-space boxes, so a ray only meets the children whose box
// it pierces.  The run's result is what the sequential visit yields: the smallest
// t, the earliest child among equal t.
// One child, with what the lanes need of its two nodes packed into 16-byte rows: inside a run
// every lane is at a different child, so these come by vector loads, not from the node array.
struct alignas(16) RunChild {
    uint32_t transform_node; // the child: a transform node (kRunNoTransform: the mesh stands in the collection as it is) ...
    uint32_t mesh_node;      // ... whose subtree ends in this mesh node
    uint32_t mesh_id;
    uint32_t material_id;    // of the mesh node
    float inverse[16];         // the transform node's matrix_inverse, row-major
    float inverse_normal[12];  // its inverse_normal_matrix, each row padded to four
    float matrix[16];          // its matrix
    float normal[12];          // its normal_matrix, rows padded
};
static_assert(sizeof(RunChild) == 240, "RunChild is read as fifteen 16-byte rows");
// Nodes in depth-first order.  A ray that pierces an inner node's box goes on to
// the next node, one that misses goes to `skip`; a pierced leaf names a child.
struct alignas(16) TlasNode {
    float bmin[3];
    uint32_t skip;
    float bmax[3];
    uint32_t child; // kTlasInner for inner nodes
};
constexpr uint32_t kTlasInner = 0xFFFFFFFFu;
constexpr uint32_t kRunNoTransform = 0xFFFFFFFFu;
struct DRun {
    const TlasNode* tlas;
    const RunChild* children; // in child order
    uint32_t tlas_count;
    uint32_t child_count;     // 0: the scene has no run
    uint32_t first_node;      // node index of the first child
    uint32_t end_node;        // one past the last child's subtree
    uint32_t only_mesh;       // the mesh every child places, 0xFFFFFFFF when they place different ones
};
// A ray's searches within the run form a chain through the job list: job_c.w of
// a job = the previous job of the same ray (index within the region), the ray's
// head = its last job.
constexpr uint32_t kRunNone = 0xFFFFFFFFu;    // end of chain / no candidate child
constexpr uint32_t kRunInPlace = 0xFFFFFFFEu; // the region's job list was full: search while walking
constexpr uint32_t kRunJobFlag = 0x80000000u; // job_c.z of a run job = flag | child index

struct DScene {
    const DNode* nodes;
    uint32_t node_count;
    const DMesh* meshes;
    const DMaterial* materials;
    uint32_t mesh_count, material_count;
    const MeshTarget* targets; // the mesh nodes outside the run, in array order
    uint32_t target_count;
    // 1: the root is a collection and each of its children is a sphere, a plane, a (kd_tree ->) mesh
    // or a transform over a (kd_tree ->) mesh -- the shape of every scene the reference ships.  The
    // walk kernel then runs one loop over the children instead of the general resumable walk.
    uint32_t flat_root;
    DRun run;
    DMesh mesh0; // copy of meshes[0]: scenes with one mesh hand it to the search kernel as an argument
};

// Camera::get_ray inputs with the per-task constants precomputed on the host.
struct DCamera {
    float res_x, res_y;      // resolution as f32
    uint32_t width, height;
    M3 rotation;
    V3 position;
    float focal_length, near_plane, bokeh_size;
    float viewport_x, viewport_y; // camera.rs:56-57
    uint32_t bokeh_shape;
};

// One record of the hit the tracer reports (RayTraceResult + bookkeeping).
struct Hit {
    float t;
    uint32_t hit, hit_inside, node_index, tri_index, material_id;
    V3 point, normal;
    V2 uv;
};

} // namespace ptk
