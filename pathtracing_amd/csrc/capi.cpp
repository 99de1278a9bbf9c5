// C ABI of the library (include/pathtracing_amd.h): scene upload, render,
// closest-hit queries, host loaders.  Nothing here may throw across the boundary.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h> // types only: the library is opened at the first multi-GPU reduce (see Rccl below)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pathtracing_amd.h"
#include "host/bvh.hpp"
#include "host/front_end.hpp"
#include "host/listener.hpp"
#include "host/host_scene.hpp"
#include "host/xform.hpp"
#include "kernels/pt_kernels.h"

namespace pt {
const std::string& last_error_string();
}

using pt::Error;

namespace {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw Error(PT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

template <class F> PtStatus guarded(F f) {
    try {
        f();
        return PT_OK;
    } catch (const Error& e) {
        pt::set_last_error(e.what());
        return e.status;
    } catch (const std::bad_alloc&) {
        pt::set_last_error("out of host memory");
        return PT_ERR_DEVICE;
    } catch (const std::exception& e) {
        pt::set_last_error(e.what());
        return PT_ERR_PARSE;
    } catch (...) {
        pt::set_last_error("unknown failure");
        return PT_ERR_INVALID_ARGUMENT;
    }
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        hip_check(hipGetDevice(&prev), "hipGetDevice");
        if (prev != dev) hip_check(hipSetDevice(dev), "hipSetDevice");
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

} // namespace

// ---- per-device workspaces -----------------------------------------------------
// The queues, job lists, result arrays and spill columns of the wavefront pipeline
// belong to the device, not to a scene: every PtScene of a device renders through the
// same (at most two) workspaces, so a worker that caches several scenes
// (worker/src/lib.rs:32-43) pays for them once.  One launch sequence uses a workspace
// at a time: `last_use` is recorded behind the last kernel that touches it, and the
// stream of the next user waits for that event, whichever scene and stream it is.
// `mu` serialises the host side (sizing and enqueueing), so calls may come from
// several threads.  A DeviceState lives until the process ends
// (pt_device_release_workspace frees its device memory earlier).
constexpr int kMaxInFlight = 4; // batches of one render traced at the same time, each in a workspace of its own
struct DeviceState {
    int device = 0;
    std::mutex mu;
    ptk::Workspace ws[kMaxInFlight];
    hipEvent_t last_use[kMaxInFlight] = {};
    bool used[kMaxInFlight] = {};
    // with batches in flight workspace k is traced on side[k], a stream of the library's own; the caller's stream
    // then only adds the batches' colours to the image, in batch order
    hipStream_t side[kMaxInFlight] = {};
    hipEvent_t fork = nullptr;                   // the caller's stream at the start of a render with batches in flight
    hipEvent_t side_done[kMaxInFlight] = {};     // a batch traced on side[k]
    hipEvent_t accumulated[kMaxInFlight] = {};   // ... and added to the image on the caller's stream
};

namespace {

constexpr int kTotals = 8; // Workspace::ray_total: rays, searches, node and triangle records fetched, wave-level step counts

DeviceState& device_state(int device) {
    static std::mutex registry_mu;
    static std::map<int, DeviceState*>* registry = new std::map<int, DeviceState*>(); // never destroyed: the HIP
    std::lock_guard<std::mutex> lock(registry_mu);                                     // runtime may be gone by then
    auto it = registry->find(device);
    if (it != registry->end()) return *it->second;
    DeviceState* ds = new DeviceState();
    ds->device = device;
    (*registry)[device] = ds;
    return *ds;
}

// The host waits until nothing queued on the device uses workspace k any more.
void wait_idle(DeviceState& ds, int k) {
    if (ds.used[k]) hip_check(hipEventSynchronize(ds.last_use[k]), "hipEventSynchronize");
    ds.used[k] = false;
}

// Whatever is queued on `stream` from here on runs after the last use of workspace k.
void stream_acquire(DeviceState& ds, int k, hipStream_t stream) {
    if (ds.used[k]) hip_check(hipStreamWaitEvent(stream, ds.last_use[k], 0), "hipStreamWaitEvent");
}

// What has been queued on `stream` so far is the last use of workspace k.
void stream_release(DeviceState& ds, int k, hipStream_t stream) {
    if (!ds.last_use[k]) hip_check(hipEventCreateWithFlags(&ds.last_use[k], hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventRecord(ds.last_use[k], stream), "hipEventRecord");
    ds.used[k] = true;
}

template <class T> void free_array(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_workspace(ptk::Workspace& ws) {
    for (int q = 0; q < 2; q++)
        for (int k = 0; k < 3; k++) free_array(ws.queue[q][k]);
    for (int k = 0; k < 3; k++) free_array(ws.jobs[k]);
    for (int k = 0; k < ptk::kMaxDeferred; k++) free_array(ws.results[k]);
    for (int k = 0; k < 2; k++) {
        free_array(ws.hits[k]);
        free_array(ws.listed[k]);
    }
    free_array(ws.hit_uv_y);
    free_array(ws.contrib);
    free_array(ws.run_head);
    free_array(ws.run_res);
    ws.capacity_paths = 0;
    ws.contrib_paths = 0;
    ws.job_records = 0;
    ws.result_arrays = 0;
}

void free_trace_state(ptk::Workspace& ws) {
    for (int q = 0; q < 2; q++) free_array(ws.wave_counts[q]);
    free_array(ws.job_counts);
    free_array(ws.ray_total);
    free_array(ws.spill);
    ws.persistent_blocks = 0;
}

// Fixed state of the trace kernels: the spill columns and per-wave queue counters of
// the persistent grid (segment_blocks_per_cu() blocks of 256 threads per compute
// unit).  Either all of it exists afterwards or none.
void ensure_trace_state(DeviceState& ds, int k) {
    ptk::Workspace& ws = ds.ws[k];
    if (ws.spill && ws.wave_counts[0] && ws.wave_counts[1] && ws.job_counts && ws.ray_total && ws.persistent_blocks) return;
    free_trace_state(ws);
    hipDeviceProp_t prop;
    hip_check(hipGetDeviceProperties(&prop, ds.device), "hipGetDeviceProperties");
    uint32_t cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    uint32_t blocks = cus * ptk::segment_blocks_per_cu();
    size_t bytes = (size_t)blocks * ptk::kTraceBlock * ptk::spill_dwords_per_thread() * sizeof(uint32_t);
    size_t waves = (size_t)blocks * (ptk::kTraceBlock / 64);
    try {
        hip_check(hipMalloc(reinterpret_cast<void**>(&ws.spill), bytes), "hipMalloc spill");
        for (int q = 0; q < 2; q++)
            hip_check(hipMalloc(reinterpret_cast<void**>(&ws.wave_counts[q]), waves * sizeof(uint32_t)), "hipMalloc counts");
        hip_check(hipMalloc(reinterpret_cast<void**>(&ws.job_counts), waves * sizeof(uint32_t)), "hipMalloc counts");
        hip_check(hipMalloc(reinterpret_cast<void**>(&ws.ray_total), kTotals * sizeof(uint64_t)), "hipMalloc ray_total");
    } catch (...) {
        free_trace_state(ws);
        throw;
    }
    ws.persistent_blocks = blocks;
}

// Queues of `records` records (all regions together), contributions of `paths` paths, job lists and
// result arrays for `result_arrays` mesh nodes searched ahead of the walk and `run_slots` job slots per
// ray for an instance run.  False: the device has not that much memory left (the workspace then holds
// nothing).  A workspace that is too small is replaced by one that covers the old and the new
// demand, so that scenes taking turns do not reallocate on every call.
bool ensure_workspace(DeviceState& ds, int k, uint64_t records, uint64_t paths, uint32_t result_arrays, uint32_t run_slots) {
    ensure_trace_state(ds, k);
    ptk::Workspace& ws = ds.ws[k];
    uint64_t job_records = records * (result_arrays + run_slots);
    const bool want_run = run_slots != 0;
    if (records <= ws.capacity_paths && paths <= ws.contrib_paths && job_records <= ws.job_records &&
        result_arrays <= ws.result_arrays && (!want_run || (ws.run_head && ws.run_res)))
        return true;
    wait_idle(ds, k);
    const uint64_t old_records = ws.capacity_paths, old_paths = ws.contrib_paths, old_jobs = ws.job_records;
    const uint32_t old_results = ws.result_arrays;
    const bool old_run = ws.run_head != nullptr;
    free_workspace(ws);
    auto attempt = [&](uint64_t rec, uint64_t pth, uint64_t jobs, uint32_t res, bool run) {
        bool ok = true;
        auto alloc = [&](auto** p, uint64_t bytes) {
            if (!ok) return;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes ? bytes : 16);
            if (e == hipErrorOutOfMemory) {
                (void)hipGetLastError(); // the failure is handled here: clear the runtime's sticky record of it
                *p = nullptr;
                ok = false;
                return;
            }
            hip_check(e, "hipMalloc workspace");
        };
        for (int q = 0; q < 2; q++)
            for (int c = 0; c < 3; c++) alloc(&ws.queue[q][c], rec * sizeof(float4));
        if (jobs)
            for (int c = 0; c < 3; c++) alloc(&ws.jobs[c], jobs * sizeof(float4));
        if (run) {
            alloc(&ws.run_head, rec * sizeof(uint32_t));
            alloc(&ws.run_res, jobs * sizeof(float4));
        }
        for (uint32_t m = 0; m < res; m++) alloc(&ws.results[m], rec * sizeof(float4));
        if (res)
            for (int q = 0; q < 2; q++) alloc(&ws.listed[q], rec * sizeof(uint32_t));
        for (int c = 0; c < 2; c++) alloc(&ws.hits[c], rec * sizeof(float4));
        alloc(&ws.hit_uv_y, rec * sizeof(float));
        alloc(&ws.contrib, pth * 3 * sizeof(float));
        if (!ok) {
            free_workspace(ws);
            return false;
        }
        ws.capacity_paths = rec;
        ws.contrib_paths = pth;
        ws.job_records = jobs;
        ws.result_arrays = res;
        return true;
    };
    const uint64_t big_rec = std::max(records, old_records), big_paths = std::max(paths, old_paths);
    const uint32_t big_res = std::max(result_arrays, old_results);
    const uint64_t big_jobs = std::max({job_records, old_jobs, big_rec * (uint64_t)(result_arrays + run_slots)});
    const bool grown = big_rec != records || big_paths != paths || big_res != result_arrays || big_jobs != job_records ||
                       (old_run && !want_run);
    if (grown && attempt(big_rec, big_paths, big_jobs, big_res, want_run || old_run)) return true;
    return attempt(records, paths, job_records, result_arrays, want_run);
}

} // namespace

// Job slots per ray set aside for the children of an instance run.  A ray of a grid of instances
// pierces a handful of boxes; regions that run out search the rest while walking.
constexpr uint32_t kRunSlotsPerRay = 8;

struct PtScene {
    int device = 0;
    ptk::DScene d{};
    std::vector<void*> allocs;
    uint64_t bvh_bytes = 0;
    uint32_t node_count = 0;
    std::vector<ptk::MeshTarget> targets; // the mesh nodes outside the instance run = BVH searches every ray needs per segment
    std::vector<uint32_t> mesh_tri_count;
    uint32_t run_slots = 0; // job slots per ray for the children of the instance run (0: no run)
    PtSceneInfo info{};

    uint32_t result_arrays() const { return std::min<uint32_t>((uint32_t)targets.size(), (uint32_t)ptk::kMaxDeferred); }
    void* dev_alloc(size_t bytes) {
        void* p = nullptr;
        hip_check(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc");
        allocs.push_back(p);
        return p;
    }
    template <class T> T* upload(const std::vector<T>& v) {
        T* p = static_cast<T*>(dev_alloc(v.size() * sizeof(T)));
        if (!v.empty()) hip_check(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
        return p;
    }
    ~PtScene() {
        for (void* p : allocs) (void)hipFree(p);
    }
};

struct PtHostScene {
    std::unique_ptr<pt::HostScene> scene;
};

namespace {

ptk::V3 v3(const PtVec3& p) { return ptk::mk3(p.x, p.y, p.z); }

// Checks the pre-order node array and fills end / parent / depth.
void layout_nodes(const PtSceneDesc& d, std::vector<ptk::DNode>& out) {
    out.assign(d.node_count, ptk::DNode{});
    struct Open {
        uint32_t node;
        uint32_t remaining;
        bool frame; // collection or transform (kd_tree is transparent)
    };
    std::vector<Open> stack;
    for (uint32_t i = 0; i < d.node_count; i++) {
        const PtNodeDesc& nd = d.nodes[i];
        if (stack.empty() && i != 0) throw Error(PT_ERR_SCENE, "node array has entries after the root subtree");
        uint32_t parent = 0xFFFFFFFFu, depth = 0;
        for (const Open& o : stack)
            if (o.frame) {
                parent = o.node;
                depth++;
            }
        ptk::DNode& o = out[i];
        o.type = nd.type;
        o.parent = parent;
        o.depth = depth;
        o.material_id = 0;
        o.mesh_id = 0;
        uint32_t children = 0;
        bool frame = false;
        switch (nd.type) {
        case PT_NODE_COLLECTION:
            children = nd.child_count;
            frame = true;
            break;
        case PT_NODE_TRANSFORM: {
            children = 1;
            frame = true;
            memcpy(o.matrix.m, nd.matrix.m, sizeof(o.matrix.m));
            o.matrix_inverse = pt::inverse4(o.matrix);
            o.normal_matrix = pt::normal_matrix4(o.matrix);
            o.inverse_normal_matrix = pt::normal_matrix4(o.matrix_inverse);
            break;
        }
        case PT_NODE_KD_TREE: children = 1; break;
        case PT_NODE_MESH:
            if (nd.mesh_id >= d.mesh_count) throw Error(PT_ERR_SCENE, "mesh node references a missing mesh");
            if (nd.material_id >= d.material_count) throw Error(PT_ERR_SCENE, "mesh node references a missing material");
            o.mesh_id = nd.mesh_id;
            o.material_id = nd.material_id;
            break;
        case PT_NODE_SPHERE:
            if (nd.material_id >= d.material_count) throw Error(PT_ERR_SCENE, "sphere references a missing material");
            o.center = v3(nd.center);
            o.radius = nd.radius;
            o.radius_sqr = nd.radius * nd.radius; // sphere.rs:51
            o.material_id = nd.material_id;
            break;
        case PT_NODE_PLANE: {
            if (nd.material_id >= d.material_count) throw Error(PT_ERR_SCENE, "plane references a missing material");
            // plane.rs:45-51
            o.point = v3(nd.point);
            o.normal = ptk::normalized(v3(nd.normal));
            o.tangent = ptk::normalized(v3(nd.tangent));
            o.bitangent = ptk::normalized(v3(nd.bitangent));
            bool ok = std::fabs(ptk::dot(o.normal, o.tangent)) < 0.001f &&
                      std::fabs(ptk::dot(o.normal, o.bitangent)) < 0.001f &&
                      std::fabs(ptk::dot(o.tangent, o.bitangent)) < 0.001f;
            if (!ok) throw Error(PT_ERR_SCENE, "plane normal/tangent/bitangent are not pairwise orthogonal");
            o.material_id = nd.material_id;
            break;
        }
        default: throw Error(PT_ERR_SCENE, "unknown node type");
        }
        if (frame && depth + 1 > (uint32_t)ptk::kMaxGraphDepth)
            throw Error(PT_ERR_UNSUPPORTED, "scene graph nests collections/transforms deeper than 8");
        // this node is one child of the innermost open node
        if (!stack.empty()) stack.back().remaining--;
        if (children > 0) {
            stack.push_back({i, children, frame});
        } else {
            o.end = i + 1;
            if (nd.type == PT_NODE_COLLECTION) o.end = i + 1; // empty collection
        }
        // close every open node whose children are all placed
        while (!stack.empty() && stack.back().remaining == 0) {
            out[stack.back().node].end = i + 1;
            stack.pop_back();
        }
    }
    if (!stack.empty()) throw Error(PT_ERR_SCENE, "node array ends inside a subtree");
}

// The mesh node a transform's subtree consists of (behind kd_tree nodes or not), or 0xFFFFFFFF.
uint32_t single_mesh_below(const std::vector<ptk::DNode>& nodes, uint32_t transform) {
    uint32_t j = transform + 1;
    while (j < nodes.size() && nodes[j].type == PT_NODE_KD_TREE) j++;
    if (j < nodes.size() && nodes[j].type == PT_NODE_MESH && nodes[transform].end == j + 1) return j;
    return 0xFFFFFFFFu;
}

struct InstanceRun {
    uint32_t first_node = 0xFFFFFFFFu, end_node = 0xFFFFFFFFu;
    std::vector<ptk::RunChild> children;
};

// A child's box in world space says something about a world-space ray only if the ray the child's
// mesh sees is the image of that ray.  Ray::apply_transform (ray.rs:21-37) maps the direction with
// the normal matrix of the inverse, i.e. with the transpose of the matrix's linear part A, and the
// points with the inverse: the two agree on the line exactly when A^T is a multiple of A^-1, that
// is A * A^T = lambda * I (rotation or reflection times one scale factor).  Under any other matrix
// the reference's ray leaves the line (a quirk kept bit for bit), and the child is visited the
// plain way.  The tolerance is far inside the widening of the boxes.
bool keeps_lines(const ptk::M4& m) {
    if (m.m[3][0] != 0.0f || m.m[3][1] != 0.0f || m.m[3][2] != 0.0f || m.m[3][3] != 1.0f) return false;
    double g[3][3], lambda = 0.0;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            g[r][c] = 0.0;
            for (int k = 0; k < 3; k++) g[r][c] += (double)m.m[r][k] * (double)m.m[c][k];
        }
    for (int r = 0; r < 3; r++) lambda += g[r][r] / 3.0;
    if (!(lambda > 1e-30) || !std::isfinite(lambda)) return false;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            if (!(std::fabs(g[r][c] - (r == c ? lambda : 0.0)) <= 1e-5 * lambda)) return false;
    for (int r = 0; r < 3; r++)
        if (!std::isfinite(m.m[r][3])) return false;
    return true;
}

// The longest run of consecutive children of the root collection that are transform -> (kd_tree ->)
// mesh subtrees, if it has at least PATHTRACING_AMD_INSTANCE_RUN_MIN children (default 4; 0 = never).
InstanceRun find_instance_run(const std::vector<ptk::DNode>& nodes) {
    InstanceRun best;
    uint32_t min_children = 4;
    if (const char* v = getenv("PATHTRACING_AMD_INSTANCE_RUN_MIN")) min_children = (uint32_t)strtoul(v, nullptr, 10);
    if (min_children == 0 || nodes.empty() || nodes[0].type != PT_NODE_COLLECTION) return best;
    InstanceRun cur;
    auto close = [&] {
        if (cur.children.size() >= min_children && cur.children.size() > best.children.size()) best = cur;
        cur = InstanceRun();
    };
    for (uint32_t c = 1; c < nodes[0].end; c = nodes[c].end) {
        // a mesh under a line-keeping transform, or a mesh as it is (behind kd_tree nodes or not)
        uint32_t mesh_node = 0xFFFFFFFFu;
        bool transformed = nodes[c].type == PT_NODE_TRANSFORM;
        if (transformed) {
            if (keeps_lines(nodes[c].matrix)) mesh_node = single_mesh_below(nodes, c);
        } else {
            uint32_t j = c;
            while (j < nodes.size() && nodes[j].type == PT_NODE_KD_TREE) j++;
            if (j < nodes.size() && nodes[j].type == PT_NODE_MESH && nodes[c].end == j + 1) mesh_node = j;
        }
        if (mesh_node == 0xFFFFFFFFu) {
            close();
            continue;
        }
        if (cur.children.empty()) cur.first_node = c;
        cur.end_node = nodes[c].end;
        ptk::RunChild rc{};
        rc.transform_node = transformed ? c : ptk::kRunNoTransform;
        rc.mesh_node = mesh_node;
        rc.mesh_id = nodes[mesh_node].mesh_id;
        rc.material_id = nodes[mesh_node].material_id;
        auto pack = [](const ptk::M4& m, const ptk::M3& n, float* m16, float* n12) {
            memcpy(m16, m.m, 16 * sizeof(float));
            const ptk::V3* rows[3] = {&n.row0, &n.row1, &n.row2};
            for (int r = 0; r < 3; r++) {
                n12[4 * r] = rows[r]->x;
                n12[4 * r + 1] = rows[r]->y;
                n12[4 * r + 2] = rows[r]->z;
                n12[4 * r + 3] = 0.0f;
            }
        };
        if (transformed) {
            pack(nodes[c].matrix_inverse, nodes[c].inverse_normal_matrix, rc.inverse, rc.inverse_normal);
            pack(nodes[c].matrix, nodes[c].normal_matrix, rc.matrix, rc.normal);
        }
        cur.children.push_back(rc);
    }
    close();
    return best;
}

// Bounding-volume tree over the world-space boxes of the run's children, in depth-first order
// with skip links.  A child's box is the box of the eight corners of its mesh's (padded) bounds
// under the child's matrix, computed in double, then widened by 1e-4 of its size and position:
// far more than the f32 rounding of the ray's way into the mesh's space and back, so a ray that
// can hit a triangle of the child pierces the box.
std::vector<ptk::TlasNode> build_run_tree(const InstanceRun& run, const std::vector<ptk::DNode>& nodes,
                                          const std::vector<ptk::DMesh>& meshes) {
    struct Box {
        double lo[3], hi[3];
    };
    const size_t n = run.children.size();
    std::vector<Box> boxes(n);
    for (size_t k = 0; k < n; k++) {
        const ptk::DMesh& mesh = meshes[run.children[k].mesh_id];
        ptk::M4 m{};
        for (int d = 0; d < 4; d++) m.m[d][d] = 1.0f;
        if (run.children[k].transform_node != ptk::kRunNoTransform) m = nodes[run.children[k].transform_node].matrix;
        Box b;
        for (int a = 0; a < 3; a++) b.lo[a] = INFINITY, b.hi[a] = -INFINITY;
        for (int corner = 0; corner < 8; corner++) {
            double p[3] = {corner & 1 ? mesh.bmax[0] : mesh.bmin[0], corner & 2 ? mesh.bmax[1] : mesh.bmin[1],
                           corner & 4 ? mesh.bmax[2] : mesh.bmin[2]};
            double w = m.m[3][0] * p[0] + m.m[3][1] * p[1] + m.m[3][2] * p[2] + m.m[3][3];
            for (int a = 0; a < 3; a++) {
                double v = (m.m[a][0] * p[0] + m.m[a][1] * p[1] + m.m[a][2] * p[2] + m.m[a][3]) / w;
                if (!(v == v)) v = INFINITY; // degenerate matrix: the box below becomes unbounded
                b.lo[a] = std::min(b.lo[a], v);
                b.hi[a] = std::max(b.hi[a], v);
            }
        }
        bool bounded = true;
        for (int a = 0; a < 3; a++) bounded = bounded && std::isfinite(b.lo[a]) && std::isfinite(b.hi[a]);
        for (int a = 0; a < 3; a++) {
            if (!bounded || mesh.tri_count == 0) { // an unbounded box is always pierced: the exact test in the child's space decides
                b.lo[a] = -INFINITY;
                b.hi[a] = INFINITY;
                continue;
            }
            double pad = 1e-4 * ((b.hi[a] - b.lo[a]) + std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]))) + 1e-6;
            b.lo[a] -= pad;
            b.hi[a] += pad;
        }
        boxes[k] = b;
    }
    std::vector<ptk::TlasNode> out;
    out.reserve(2 * n);
    auto down = [](double v) { return std::nextafterf((float)v, -INFINITY); };
    auto up = [](double v) { return std::nextafterf((float)v, INFINITY); };
    // children are put into the tree in any order: the run's result does not depend on the order of visit
    std::vector<uint32_t> order(n);
    for (size_t k = 0; k < n; k++) order[k] = (uint32_t)k;
    struct Builder {
        std::vector<ptk::TlasNode>& out;
        std::vector<Box>& boxes;
        std::vector<uint32_t>& order;
        decltype(down)& dn;
        decltype(up)& upf;
        Box build(size_t lo, size_t hi) {
            size_t at = out.size();
            out.push_back(ptk::TlasNode{});
            Box b;
            if (hi - lo == 1) {
                b = boxes[order[lo]];
                out[at].child = order[lo];
            } else {
                double cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
                auto centre = [&](uint32_t k, int a) {
                    double c = 0.5 * (boxes[k].lo[a] + boxes[k].hi[a]);
                    return std::isfinite(c) ? c : 0.0;
                };
                for (size_t i = lo; i < hi; i++)
                    for (int a = 0; a < 3; a++) {
                        cmin[a] = std::min(cmin[a], centre(order[i], a));
                        cmax[a] = std::max(cmax[a], centre(order[i], a));
                    }
                int axis = 0;
                for (int a = 1; a < 3; a++)
                    if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
                size_t mid = lo + (hi - lo) / 2;
                std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                                 [&](uint32_t x, uint32_t y) { return centre(x, axis) < centre(y, axis); });
                Box l = build(lo, mid), r = build(mid, hi);
                for (int a = 0; a < 3; a++) {
                    b.lo[a] = std::min(l.lo[a], r.lo[a]);
                    b.hi[a] = std::max(l.hi[a], r.hi[a]);
                }
                out[at].child = ptk::kTlasInner;
            }
            for (int a = 0; a < 3; a++) {
                out[at].bmin[a] = dn(b.lo[a]);
                out[at].bmax[a] = upf(b.hi[a]);
            }
            out[at].skip = (uint32_t)out.size();
            return b;
        }
    } builder{out, boxes, order, down, up};
    builder.build(0, n);
    for (size_t i = 0; i < out.size(); i++)
        if (out[i].skip <= i) throw Error(PT_ERR_SCENE, "instance tree: skip link does not advance");
    return out;
}

void build_mesh(PtScene& s, const PtMeshDesc& m, const PtTreeBlob* blob, ptk::DMesh* out) {
    if (m.triangle_count && !m.positions) throw Error(PT_ERR_INVALID_ARGUMENT, "mesh without positions");
    if (m.triangle_count >= (1ull << 28)) throw Error(PT_ERR_UNSUPPORTED, "mesh has too many triangles");
    const uint64_t n = m.triangle_count;
    // A .tree blob is adopted only if it describes exactly this mesh and the kernels can search it;
    // anything else (empty, stale, corrupt, too deep) is ignored and the tree is built here.
    pt::BuiltBvh bvh;
    pt::WideBvh wide;
    bool adopted = false;
    const bool offered = blob && blob->size && blob->data;
    if (offered) {
        try {
            if (pt::deserialize_tree(blob->data, blob->size, m.positions, n, &bvh)) {
                wide = pt::widen_bvh(bvh);
                adopted = wide.max_stack <= ptk::traversal_stack_capacity();
            }
        } catch (const Error&) {
            adopted = false;
        }
    }
    if (!adopted) {
        auto t0 = std::chrono::steady_clock::now();
        bvh = pt::build_bvh(m.positions, n, pt::default_bvh_tuning().max_leaf);
        wide = pt::widen_bvh(bvh);
        s.info.build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (wide.max_stack > ptk::traversal_stack_capacity())
            throw Error(PT_ERR_UNSUPPORTED, "BVH needs a deeper traversal stack than the kernels keep");
    }
    s.info.trees_offered += offered ? 1u : 0u;
    s.info.trees_adopted += adopted ? 1u : 0u;
    s.info.trees_built += adopted ? 0u : 1u;
    s.info.triangle_count += n;

    std::vector<ptk::LeafTri> leaf(n), orig(n);
    std::vector<ptk::ShadeTri> shade(n);
    for (uint64_t i = 0; i < n; i++) {
        const float* p = m.positions + 9 * i;
        ptk::LeafTri& t = orig[i];
        memcpy(t.v0, p, 12);
        memcpy(t.v1, p + 3, 12);
        memcpy(t.v2, p + 6, 12);
        t.orig_index = (uint32_t)i;
        t.pad1 = t.pad2 = 0;
        // TriangleUninit::init, triangle/mod.rs:65-84
        ptk::V3 v0 = ptk::ld3(p), v1 = ptk::ld3(p + 3), v2 = ptk::ld3(p + 6);
        ptk::V3 tn = ptk::normalized(ptk::cross(v1 - v0, v1 - v2));
        ptk::ShadeTri& sh = shade[i];
        memset(&sh, 0, sizeof(sh));
        sh.true_normal[0] = tn.x;
        sh.true_normal[1] = tn.y;
        sh.true_normal[2] = tn.z;
        bool has_n = m.normals && (!m.normal_valid || m.normal_valid[i]);
        float* dstn[3] = {sh.n0, sh.n1, sh.n2};
        float* dstuv[3] = {sh.uv0, sh.uv1, sh.uv2};
        for (int v = 0; v < 3; v++) {
            if (has_n) memcpy(dstn[v], m.normals + 9 * i + 3 * v, 12);
            else memcpy(dstn[v], sh.true_normal, 12);
            if (m.uvs) memcpy(dstuv[v], m.uvs + 6 * i + 2 * v, 8);
        }
    }
    for (uint64_t k = 0; k < n; k++) leaf[k] = orig[bvh.order[k]];

    out->nodes = s.upload(wide.nodes);
    out->tris = s.upload(leaf);
    out->tris_orig = s.upload(orig);
    out->shade = s.upload(shade);
    out->tri_count = (uint32_t)n;
    out->node_count = (uint32_t)wide.nodes.size();
    out->root_ref = wide.root_ref;
    out->pad = 0;
    memcpy(out->bmin, bvh.bmin, sizeof(out->bmin));
    memcpy(out->bmax, bvh.bmax, sizeof(out->bmax));
    s.bvh_bytes += wide.nodes.size() * sizeof(ptk::WideNode) + n * sizeof(ptk::LeafTri);
}

ptk::DCamera make_camera(const PtCamera& c) {
    if (c.resolution_x == 0 || c.resolution_y == 0) throw Error(PT_ERR_INVALID_ARGUMENT, "camera resolution is zero");
    if (c.resolution_x > 65536 || c.resolution_y > 65536) throw Error(PT_ERR_UNSUPPORTED, "camera resolution above 65536");
    if (c.bokeh_shape > PT_BOKEH_SQUARE) throw Error(PT_ERR_INVALID_ARGUMENT, "unknown bokeh shape");
    ptk::DCamera d;
    d.width = (uint32_t)c.resolution_x;
    d.height = (uint32_t)c.resolution_y;
    d.res_x = (float)c.resolution_x;
    d.res_y = (float)c.resolution_y;
    d.rotation.row0 = v3(c.rotation.row0);
    d.rotation.row1 = v3(c.rotation.row1);
    d.rotation.row2 = v3(c.rotation.row2);
    d.position = v3(c.position);
    d.focal_length = c.focal_length;
    d.near_plane = c.near_plane;
    d.bokeh_size = c.bokeh_size;
    d.bokeh_shape = c.bokeh_shape;
    // camera.rs:56-57
    d.viewport_x = c.focal_length * tanf(c.fov * 0.5f) * 2.0f;
    d.viewport_y = d.viewport_x * (d.res_y / d.res_x);
    return d;
}

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { hip_check(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc"); }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

} // namespace

// client/src/frame.rs:11-15,54-58: the running sum of the renders of one frame and
// how many there were, kept on one device.
struct PtFrame {
    int device = 0;
    uint32_t width = 0, height = 0;
    uint64_t count = 0;
    float* sum = nullptr;     // width * height * 3
    float* staging = nullptr; // one render uploaded from the host; the mean image on its way back
    uint8_t* rgba = nullptr;  // width * height * 4
    size_t floats() const { return (size_t)width * height * 3; }
    ~PtFrame() {
        if (sum) (void)hipFree(sum);
        if (staging) (void)hipFree(staging);
        if (rgba) (void)hipFree(rgba);
    }
};

namespace {

// How many batches of a render may be traced at the same time: PATHTRACING_AMD_STREAMS (1..4), default 2.
int batches_in_flight_allowed() {
    const char* v = getenv("PATHTRACING_AMD_STREAMS");
    int n = v ? atoi(v) : 2;
    return std::max(1, std::min(n, kMaxInFlight));
}

void render_accumulate(PtScene* scene, const PtRenderTask* task, const PtRenderOpts* opts_in, float* d_accum,
                       hipStream_t stream, PtRenderStats* stats) {
    if (!scene || !task || !d_accum) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
    PtRenderOpts o;
    memset(&o, 0, sizeof(o));
    if (opts_in) o = *opts_in;
    ptk::DCamera cam = make_camera(task->camera);
    if (task->trace_depth > 0xFFFFFFFEull) throw Error(PT_ERR_UNSUPPORTED, "trace_depth above 2^32");
    if (o.sample_offset + task->iterations > 0xFFFFFFFFull) throw Error(PT_ERR_UNSUPPORTED, "sample index above 2^32");
    uint32_t x0 = 0, y0 = 0, w = cam.width, h = cam.height;
    if (o.tile_w && o.tile_h) {
        if (o.tile_x0 >= cam.width || o.tile_y0 >= cam.height) throw Error(PT_ERR_INVALID_ARGUMENT, "tile outside the frame");
        x0 = o.tile_x0;
        y0 = o.tile_y0;
        w = std::min(o.tile_w, cam.width - x0);
        h = std::min(o.tile_h, cam.height - y0);
    }
    DeviceState& ds = device_state(scene->device);
    std::lock_guard<std::mutex> lock(ds.mu);
    const bool brute = (o.flags & PT_FLAG_BRUTE_FORCE) != 0;
    const bool profile = stats && (o.flags & PT_FLAG_PROFILE) != 0;
    const bool count_visits = stats && (o.flags & PT_FLAG_COUNT_VISITS) != 0;
    // Path slots per batch.  Bigger batches are faster (fewer launches whose last waves run alone, more
    // samples of a pixel block next to each other), and two batches in flight are faster than one of
    // twice the size (what one batch's launches leave idle, the other's fill: 96 ms against 107 per
    // 1080p x 64 spp frame of the 871k-triangle scene).  A slot costs about 210 bytes of queues, nine times
    // the job records on top for scenes with an instance run.  The default is halved until it fits.
    uint64_t budget = o.paths_per_batch ? o.paths_per_batch : ((scene->run_slots ? 16ull : 64ull) << 20);
    if (budget < 64) budget = 64;
    const uint32_t nbx = (w + 7) / 8;
    const uint64_t slots_per_block_row = (uint64_t)nbx * 64;
    const uint32_t total_block_rows = (h + 7) / 8;
    uint32_t band_block_rows = 0, samples_per_batch = 0;
    int flight = 1; // batches in flight: batch i is traced in workspace i mod flight (flight > 1: on side stream i mod flight)
    ensure_trace_state(ds, 0);
    const uint32_t waves = ds.ws[0].persistent_blocks * (ptk::kTraceBlock / 64);
    for (;;) {
        // bands of whole 8-pixel rows such that one sample of a band fits the budget
        band_block_rows = (uint32_t)std::min<uint64_t>(total_block_rows, std::max<uint64_t>(1, budget / slots_per_block_row));
        uint64_t band_slots = (uint64_t)band_block_rows * slots_per_block_row;
        samples_per_batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(budget / band_slots, task->iterations ? task->iterations : 1));
        // a batch indexes its path slots with 32 bits
        while ((uint64_t)samples_per_batch * band_slots > 0xFFFFFFFFull && samples_per_batch > 1) samples_per_batch /= 2;
        flight = (profile || count_visits || (o.flags & PT_FLAG_SINGLE_STREAM)) ? 1 : batches_in_flight_allowed();
        // The samples go into batches of equal size (64 as 22 + 22 + 20, not 30 + 30 + 4), and a render that one
        // batch would hold is cut into two when two may be in flight and each half is still a big launch: two
        // batches of 8 Mi paths side by side beat one of 16 Mi by 6 %, two of 16 Mi one of 32 Mi by 9 %, two of 32 Mi one
        // of 64 Mi by 11 %; below 8 Mi in all it makes no difference (tools/probe_samples.py).  This is what a rank of a
        // frame shared among GPUs by sample ranges renders.
        {
            uint64_t sample_batches = (task->iterations + samples_per_batch - 1) / samples_per_batch;
            const bool one_band = band_block_rows >= total_block_rows;
            if (sample_batches == 1 && one_band && flight > 1 && !o.paths_per_batch && task->iterations >= 2 &&
                band_slots * task->iterations >= (8ull << 20))
                sample_batches = 2;
            if (sample_batches > 1)
                samples_per_batch = (uint32_t)((task->iterations + sample_batches - 1) / sample_batches);
        }
        const uint64_t n_batches = (uint64_t)((total_block_rows + band_block_rows - 1) / band_block_rows) *
                                   ((task->iterations + samples_per_batch - 1) / samples_per_batch);
        flight = (int)std::min<uint64_t>((uint64_t)flight, std::max<uint64_t>(n_batches, 1));
        const uint64_t records = (uint64_t)ptk::queue_region_capacity(nbx, band_block_rows, samples_per_batch, waves) * waves;
        const uint64_t paths = band_slots * samples_per_batch;
        bool ok = ensure_workspace(ds, 0, records, paths, scene->result_arrays(), scene->run_slots);
        for (int k = 1; ok && k < flight; k++)
            if (!ensure_workspace(ds, k, records, paths, scene->result_arrays(), scene->run_slots)) flight = k; // what fits
        if (ok) break;
        if (o.paths_per_batch || budget <= (1ull << 20))
            throw Error(PT_ERR_DEVICE, "out of device memory for the render's queues (paths_per_batch " + std::to_string(budget) + ")");
        budget /= 2;
    }
    const bool piped = flight > 1;
    for (int k = 0; piped && k < flight; k++) {
        if (!ds.side[k]) hip_check(hipStreamCreateWithFlags(&ds.side[k], hipStreamNonBlocking), "hipStreamCreate");
        for (hipEvent_t* e : {&ds.fork, &ds.side_done[k], &ds.accumulated[k]})
            if (!*e) hip_check(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreate");
    }

    std::vector<hipEvent_t> events; // [0], [1]: around the render (stats); then pairs around search launches (profile)
    struct EventCleanup {
        std::vector<hipEvent_t>& v;
        ~EventCleanup() {
            for (hipEvent_t e : v) (void)hipEventDestroy(e);
        }
    } cleanup{events};
    std::vector<hipEvent_t> seg_events;
    EventCleanup seg_cleanup{seg_events};
    // A failure while enqueueing leaves launches behind that no event covers: wait for the device
    // before the error travels on, so that the next call finds the workspaces idle.
    struct DrainOnFailure {
        bool armed = true;
        ~DrainOnFailure() {
            if (armed) (void)hipDeviceSynchronize();
        }
    } drain;
    // whatever used the workspaces last (another scene, another stream) comes first
    if (!piped) stream_acquire(ds, 0, stream);
    if (piped) hip_check(hipEventRecord(ds.fork, stream), "hipEventRecord");
    for (int k = 0; piped && k < flight; k++) {
        hip_check(hipStreamWaitEvent(ds.side[k], ds.fork, 0), "hipStreamWaitEvent");
        stream_acquire(ds, k, ds.side[k]);
    }
    if (stats) {
        hipEvent_t e = nullptr;
        hip_check(hipEventCreate(&e), "hipEventCreate");
        events.push_back(e);
        hip_check(hipEventCreate(&e), "hipEventCreate");
        events.push_back(e);
        for (int k = 0; k < flight; k++)
            hip_check(hipMemsetAsync(ds.ws[k].ray_total, 0, kTotals * sizeof(uint64_t), piped ? ds.side[k] : stream), "hipMemsetAsync");
        hip_check(hipEventRecord(events[0], stream), "hipEventRecord");
    }
    uint64_t paths = 0, batch_index = 0;
    bool holds_contrib[kMaxInFlight] = {}; // workspace k's colours are still to be read by an accumulate on `stream`
    for (uint32_t br = 0; br < total_block_rows; br += band_block_rows) {
        ptk::BatchGeometry g;
        g.x0 = x0;
        g.y0 = y0 + br * 8;
        g.w = w;
        g.h = std::min<uint32_t>(band_block_rows * 8, h - br * 8);
        g.nbx = nbx;
        g.nby = (g.h + 7) / 8;
        g.npp = g.nbx * g.nby * 64;
        for (uint64_t s = 0; s < task->iterations; s += samples_per_batch, batch_index++) {
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(samples_per_batch, task->iterations - s);
            const int k = (int)(batch_index % (uint64_t)flight);
            hipStream_t st = piped ? ds.side[k] : stream;
            // a workspace is free again once the colours of its previous batch have been added to the image
            if (piped && holds_contrib[k]) hip_check(hipStreamWaitEvent(ds.side[k], ds.accumulated[k], 0), "hipStreamWaitEvent");
            hip_check(ptk::render_batch(scene->d, cam, g, o.seed, o.sample_offset + s, cnt, (uint32_t)task->trace_depth,
                                        brute, scene->targets, scene->run_slots, ds.ws[k], st,
                                        profile ? &seg_events : nullptr, count_visits),
                      "render_batch");
            if (piped) {
                hip_check(hipEventRecord(ds.side_done[k], ds.side[k]), "hipEventRecord");
                hip_check(hipStreamWaitEvent(stream, ds.side_done[k], 0), "hipStreamWaitEvent");
            }
            // the per-pixel sums stay in sample order: every batch is added on the caller's stream, in batch order
            hip_check(ptk::accumulate_batch(g, cam.width, cnt, ds.ws[k], d_accum, stream), "accumulate_batch");
            if (piped) {
                hip_check(hipEventRecord(ds.accumulated[k], stream), "hipEventRecord");
                holds_contrib[k] = true;
            }
            paths += (uint64_t)cnt * g.w * g.h;
        }
    }
    for (int k = 0; k < flight; k++) stream_release(ds, k, stream); // `stream` has waited for everything queued on the side streams
    drain.armed = false;
    if (stats) {
        hip_check(hipEventRecord(events[1], stream), "hipEventRecord");
        hip_check(hipEventSynchronize(events[1]), "hipEventSynchronize");
        float ms = 0.0f;
        hip_check(hipEventElapsedTime(&ms, events[0], events[1]), "hipEventElapsedTime");
        uint64_t totals[kTotals] = {};
        for (int k = 0; k < flight; k++) {
            uint64_t part[kTotals] = {};
            hip_check(hipMemcpy(part, ds.ws[k].ray_total, sizeof(part), hipMemcpyDeviceToHost), "hipMemcpy D2H");
            for (int c = 0; c < kTotals; c++) totals[c] += part[c];
        }
        stats->rays = task->trace_depth ? totals[0] : 0;
        stats->paths = paths;
        stats->render_ms = ms;
        stats->bvh_bytes = scene->bvh_bytes;
        stats->searches = totals[1];
        stats->node_visits = totals[2];
        stats->triangle_tests = totals[3];
        stats->wave_node_steps = totals[4];
        stats->wave_leaf_steps = totals[5];
        stats->wave_leaf_rounds = totals[6];
        stats->search_kernel_ms = 0.0;
        stats->search_kernel_launches = 0;
        for (size_t i = 0; i + 1 < seg_events.size(); i += 2) {
            float seg = 0.0f;
            hip_check(hipEventElapsedTime(&seg, seg_events[i], seg_events[i + 1]), "hipEventElapsedTime");
            stats->search_kernel_ms += seg;
            stats->search_kernel_launches++;
        }
    }
}

} // namespace


// ---------------------------------------------------------------------- .tree cache, worker
namespace {

// The write side of the .tree cache: every kd_tree node directly above a mesh node whose file is missing, empty
// or not valid for the mesh gets its tree built, written through `store` and taken into the loaded scene.
uint32_t cache_trees(pt::HostScene& hs, const pt::FileStore& store) {
    uint32_t written = 0;
    // the description's blob table must point at the new contents however this function is left: a write that
    // fails part-way (PT_ERR_IO, which callers may ignore) comes after earlier blobs have been replaced
    struct Finalize {
        pt::HostScene& hs;
        ~Finalize() { hs.finalize(); }
    } finalize{hs};
    std::vector<uint8_t> settled(hs.trees.size(), 0); // this call has dealt with the file (one mesh per file)
    for (size_t i = 0; i + 1 < hs.nodes.size(); i++) {
        const PtNodeDesc& nd = hs.nodes[i];
        if (nd.type != PT_NODE_KD_TREE || hs.nodes[i + 1].type != PT_NODE_MESH || nd.tree_id == PT_NO_INDEX) continue;
        const uint32_t t = nd.tree_id, m = hs.nodes[i + 1].mesh_id;
        if (t >= hs.trees.size() || m >= hs.meshes.size() || settled[t]) continue;
        settled[t] = 1;
        const pt::HostMesh& mesh = hs.meshes[m];
        if (mesh.triangle_count == 0) continue;
        pt::BuiltBvh bvh;
        if (!hs.trees[t].empty() &&
            pt::deserialize_tree(hs.trees[t].data(), hs.trees[t].size(), mesh.positions.data(), mesh.triangle_count, &bvh))
            continue; // a hit: nothing to write
        bvh = pt::build_bvh(mesh.positions.data(), mesh.triangle_count, pt::default_bvh_tuning().max_leaf);
        std::vector<uint8_t> blob = pt::serialize_tree(bvh, mesh.positions.data(), mesh.triangle_count);
        store.write_file(hs.tree_paths[t], blob);
        hs.trees[t] = std::move(blob);
        written++;
    }
    return written;
}

} // namespace

// Worker (worker/src/lib.rs:18-49) with the FileStore (file_store.rs:12-35) over a directory: the bucket of
// a scene is the sub-directory named by its scene_md5.
struct PtWorker {
    std::string store_root;
    int device = 0;
    std::mutex mu; // the reference keeps its Worker behind a mutex held across the render (main.rs:32,82)
    struct Entry {
        std::unique_ptr<pt::HostScene> host;
        PtScene* scene = nullptr;
    };
    std::map<std::string, Entry> cached_scenes; // lib.rs:20: keyed by scene_md5
    ~PtWorker() {
        for (auto& kv : cached_scenes) pt_scene_destroy(kv.second.scene);
    }
};

namespace {

// ------------------------------------------------------------ several GPUs, one process

// The share of `rank` of `world` (the split pathtracing_amd/distributed.py makes for the one-process-per-GPU path).
// Sample ranges are contiguous and disjoint, the first `iterations % world` ranks get one sample more; row bands are
// whole 8-pixel block rows except the last.  False: nothing to render.
bool shard(const PtRenderTask& task, const PtRenderOpts& opts, uint32_t mode, uint32_t world, uint32_t rank,
           PtRenderTask* out_task, PtRenderOpts* out_opts) {
    if (world == 0 || rank >= world) throw Error(PT_ERR_INVALID_ARGUMENT, "bad rank / world size");
    *out_task = task;
    *out_opts = opts;
    if (mode == PT_SHARD_SAMPLES) {
        const uint64_t base = task.iterations / world, extra = task.iterations % world;
        const uint64_t count = base + (rank < extra ? 1 : 0);
        const uint64_t first = (uint64_t)rank * base + std::min<uint64_t>(rank, extra);
        out_task->iterations = count;
        out_opts->sample_offset = opts.sample_offset + first;
        return count != 0;
    }
    if (mode == PT_SHARD_ROWS) {
        const uint64_t height = task.camera.resolution_y, width = task.camera.resolution_x;
        const uint64_t blocks = (height + 7) / 8, base = blocks / world, extra = blocks % world;
        const uint64_t nblk = base + (rank < extra ? 1 : 0);
        const uint64_t first_blk = (uint64_t)rank * base + std::min<uint64_t>(rank, extra);
        const uint64_t y0 = first_blk * 8;
        const uint64_t end = std::min<uint64_t>(height, (first_blk + nblk) * 8);
        const uint64_t rows = end > y0 ? end - y0 : 0;
        out_opts->tile_x0 = 0;
        out_opts->tile_y0 = (uint32_t)y0;
        out_opts->tile_w = (uint32_t)width;
        out_opts->tile_h = (uint32_t)rows;
        return rows != 0;
    }
    throw Error(PT_ERR_INVALID_ARGUMENT, "unknown sharding mode");
}

// RCCL, opened on first use by its soname: a process that has loaded another copy already (PyTorch wheels bundle
// one under the same soname) keeps using that copy, and a single-GPU process never loads it.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
    Rccl() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return;
        comm_init_all = reinterpret_cast<decltype(comm_init_all)>(dlsym(handle, "ncclCommInitAll"));
        comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(handle, "ncclCommDestroy"));
        reduce = reinterpret_cast<decltype(reduce)>(dlsym(handle, "ncclReduce"));
        group_start = reinterpret_cast<decltype(group_start)>(dlsym(handle, "ncclGroupStart"));
        group_end = reinterpret_cast<decltype(group_end)>(dlsym(handle, "ncclGroupEnd"));
        error_string = reinterpret_cast<decltype(error_string)>(dlsym(handle, "ncclGetErrorString"));
        ok = comm_init_all && comm_destroy && reduce && group_start && group_end && error_string;
    }
    void check(ncclResult_t r, const char* what) const {
        if (r != ncclSuccess) throw Error(PT_ERR_DEVICE, std::string(what) + ": " + error_string(r));
    }
};
Rccl& rccl() {
    static Rccl* r = new Rccl(); // never unloaded
    return *r;
}

// What pt_render_multi keeps between calls for one list of devices: a stream and an accumulation buffer per
// entry of the list, the staging buffer of the peer-copy route and the resolved image on the first device.
// Calls on the same list take turns (the buffers, and the communicators' order of operations, are shared).
struct MultiSlot {
    int device = 0;
    hipStream_t stream = nullptr;
    float* accum = nullptr;
};
struct MultiContext {
    std::mutex mu;
    std::vector<MultiSlot> slots;
    size_t capacity = 0; // bytes of every buffer below
    float* staging = nullptr;
    float* resolved = nullptr;
    std::vector<ncclComm_t> comms; // empty until the RCCL route is first taken
};
std::mutex& multi_registry_mu() {
    static std::mutex mu;
    return mu;
}
std::map<std::vector<int>, MultiContext*>& multi_registry() {
    static auto* r = new std::map<std::vector<int>, MultiContext*>(); // never destroyed: the HIP runtime may be gone by then
    return *r;
}
MultiContext& multi_context(const std::vector<int>& devices) {
    std::lock_guard<std::mutex> lock(multi_registry_mu());
    MultiContext*& c = multi_registry()[devices];
    if (!c) {
        c = new MultiContext();
        c->slots.resize(devices.size());
        for (size_t i = 0; i < devices.size(); i++) c->slots[i].device = devices[i];
    }
    return *c;
}
void free_multi_buffers(MultiContext& c) {
    for (MultiSlot& s : c.slots) {
        if (!s.accum) continue;
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        (void)hipFree(s.accum);
        s.accum = nullptr;
    }
    if (!c.slots.empty()) (void)hipSetDevice(c.slots[0].device);
    if (c.staging) (void)hipFree(c.staging);
    if (c.resolved) (void)hipFree(c.resolved);
    c.staging = c.resolved = nullptr;
    c.capacity = 0;
}
// Frees the buffers of every device list that names `device` (streams and communicators stay).
void release_multi_buffers(int device) {
    std::vector<MultiContext*> hit;
    {
        std::lock_guard<std::mutex> lock(multi_registry_mu());
        for (auto& kv : multi_registry())
            if (std::find(kv.first.begin(), kv.first.end(), device) != kv.first.end()) hit.push_back(kv.second);
    }
    for (MultiContext* c : hit) {
        std::lock_guard<std::mutex> lock(c->mu);
        free_multi_buffers(*c);
    }
}
void prepare_multi(MultiContext& c, size_t bytes, bool want_staging) {
    if (bytes > c.capacity) free_multi_buffers(c);
    const size_t cap = std::max<size_t>(std::max(bytes, c.capacity), 16);
    for (MultiSlot& s : c.slots) {
        hip_check(hipSetDevice(s.device), "hipSetDevice");
        if (!s.stream) hip_check(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreate");
        if (!s.accum) hip_check(hipMalloc(reinterpret_cast<void**>(&s.accum), cap), "hipMalloc accumulation buffer");
    }
    hip_check(hipSetDevice(c.slots[0].device), "hipSetDevice");
    if (!c.resolved) hip_check(hipMalloc(reinterpret_cast<void**>(&c.resolved), cap), "hipMalloc image");
    if (want_staging && !c.staging) hip_check(hipMalloc(reinterpret_cast<void**>(&c.staging), cap), "hipMalloc staging");
    c.capacity = cap;
}

// One device's part of a pt_render_multi call.
struct Lane {
    PtScene* scene = nullptr;
    PtRenderTask task{};
    PtRenderOpts opts{};
    bool active = false;
    PtRenderStats stats{};
    PtStatus status = PT_OK;
    std::string message;
};

void render_multi(PtScene** scenes, int n, const PtRenderTask* task, const PtRenderOpts* opts_in, uint32_t mode,
                  float* out_rgb, PtRenderStats* stats) {
    if (!scenes || !task || !out_rgb) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 1 || n > 64) throw Error(PT_ERR_INVALID_ARGUMENT, "pt_render_multi takes 1 to 64 scenes");
    for (int i = 0; i < n; i++)
        if (!scenes[i]) throw Error(PT_ERR_INVALID_ARGUMENT, "null scene");
    (void)make_camera(task->camera);
    // work_group.rs:67-69 scales by 1 / iterations: without a sample there is no image to scale
    if (task->iterations == 0) throw Error(PT_ERR_INVALID_ARGUMENT, "pt_render_multi needs at least one iteration");
    PtRenderOpts base;
    memset(&base, 0, sizeof(base));
    if (opts_in) base = *opts_in;
    if (mode == PT_SHARD_ROWS && (base.tile_w != 0 || base.tile_h != 0))
        throw Error(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: a tile cannot be combined with row bands (the bands are the tiles)");
    const size_t floats = (size_t)task->camera.resolution_x * task->camera.resolution_y * 3;
    const size_t bytes = floats * sizeof(float);
    int prev_device = 0;
    hip_check(hipGetDevice(&prev_device), "hipGetDevice");
    struct Restore {
        int dev;
        ~Restore() { (void)hipSetDevice(dev); }
    } restore{prev_device};

    std::vector<int> devices;
    bool distinct = true;
    for (int i = 0; i < n; i++) {
        for (int d : devices) distinct = distinct && d != scenes[i]->device;
        devices.push_back(scenes[i]->device);
    }
    const char* how = getenv("PATHTRACING_AMD_REDUCE");
    const bool want_p2p = how && strcmp(how, "p2p") == 0;
    const bool use_rccl = n > 1 && distinct && !want_p2p && rccl().ok;

    // set-up, outside the time reported: cached buffers and streams, and the communicators on their first use
    MultiContext& ctx = multi_context(devices);
    std::lock_guard<std::mutex> turn(ctx.mu);
    prepare_multi(ctx, bytes, n > 1 && !use_rccl);
    if (use_rccl && ctx.comms.empty()) {
        std::vector<ncclComm_t> comms(devices.size());
        rccl().check(rccl().comm_init_all(comms.data(), (int)devices.size(), devices.data()), "ncclCommInitAll");
        ctx.comms = comms; // kept for the life of the process: setting one up takes seconds
    }
    // where the image goes: host memory, or memory of the first device (then no copy is made)
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    const bool out_on_device = hipPointerGetAttributes(&attr, out_rgb) == hipSuccess && attr.type == hipMemoryTypeDevice;
    if (!out_on_device) (void)hipGetLastError(); // an unregistered host pointer is reported as an error: not one here
    if (out_on_device && attr.device != devices[0])
        throw Error(PT_ERR_INVALID_ARGUMENT, "pt_render_multi: a device image buffer must live on the first scene's device");

    const auto t_start = std::chrono::steady_clock::now();
    std::vector<Lane> lanes((size_t)n);
    // whatever happens below, nothing may still be running on the shared buffers when the turn ends
    struct Drain {
        MultiContext& c;
        ~Drain() {
            for (MultiSlot& s : c.slots) {
                (void)hipSetDevice(s.device);
                (void)hipStreamSynchronize(s.stream);
            }
        }
    } drain{ctx};
    for (int i = 0; i < n; i++) {
        Lane& l = lanes[(size_t)i];
        l.scene = scenes[i];
        l.active = shard(*task, base, mode, (uint32_t)n, (uint32_t)i, &l.task, &l.opts);
        hip_check(hipSetDevice(l.scene->device), "hipSetDevice");
        hip_check(hipMemsetAsync(ctx.slots[(size_t)i].accum, 0, bytes, ctx.slots[(size_t)i].stream), "hipMemsetAsync");
    }
    // every device renders its share; one host thread per device keeps them all busy
    std::vector<std::thread> threads;
    for (int i = 0; i < n; i++) {
        Lane* l = &lanes[(size_t)i];
        MultiSlot* s = &ctx.slots[(size_t)i];
        if (!l->active) continue;
        threads.emplace_back([l, s] {
            l->status = guarded([&] {
                hip_check(hipSetDevice(l->scene->device), "hipSetDevice");
                render_accumulate(l->scene, &l->task, &l->opts, s->accum, s->stream, &l->stats);
            });
            if (l->status != PT_OK) l->message = pt::last_error_string();
        });
    }
    for (std::thread& t : threads) t.join();
    for (const Lane& l : lanes)
        if (l.status != PT_OK) throw Error(l.status, l.message);
    for (MultiSlot& s : ctx.slots) { // lanes without a share have only their memset queued
        hip_check(hipSetDevice(s.device), "hipSetDevice");
        hip_check(hipStreamSynchronize(s.stream), "hipStreamSynchronize");
    }

    // the sum of the un-normalised buffers onto the first device (client/src/frame.rs:60-71 does it on the host)
    MultiSlot& root = ctx.slots[0];
    const auto t_reduce = std::chrono::steady_clock::now();
    if (n > 1) {
        if (use_rccl) {
            // RCCL over xGMI: one grouped ncclReduce(sum) with root 0, each rank on its own stream
            rccl().check(rccl().group_start(), "ncclGroupStart");
            for (int i = 0; i < n; i++) {
                MultiSlot& s = ctx.slots[(size_t)i];
                hip_check(hipSetDevice(s.device), "hipSetDevice");
                rccl().check(rccl().reduce(s.accum, root.accum, floats, ncclFloat, ncclSum, 0, ctx.comms[(size_t)i], s.stream),
                             "ncclReduce");
            }
            rccl().check(rccl().group_end(), "ncclGroupEnd");
            for (MultiSlot& s : ctx.slots) {
                hip_check(hipSetDevice(s.device), "hipSetDevice");
                hip_check(hipStreamSynchronize(s.stream), "hipStreamSynchronize");
            }
        } else {
            // peer copies: every other buffer is copied to the first device and added there, in device order
            // (also the route for several scenes on one device, which RCCL does not take)
            hip_check(hipSetDevice(root.device), "hipSetDevice");
            for (int i = 1; i < n; i++) {
                hip_check(hipMemcpyPeerAsync(ctx.staging, root.device, ctx.slots[(size_t)i].accum, ctx.slots[(size_t)i].device, bytes,
                                             root.stream),
                          "hipMemcpyPeerAsync");
                hip_check(ptk::frame_add(root.accum, ctx.staging, floats, root.stream), "frame_add");
            }
            hip_check(hipStreamSynchronize(root.stream), "hipStreamSynchronize");
        }
    }
    const double reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_reduce).count();
    hip_check(hipSetDevice(root.device), "hipSetDevice");
    // get_image: work_group.rs:67-69 scale, cpu_renderer/mod.rs:143-175 flip
    const float scale = 1.0f / (float)task->iterations;
    float* image = out_on_device ? out_rgb : ctx.resolved;
    hip_check(ptk::resolve_image(root.accum, (uint32_t)task->camera.resolution_x, (uint32_t)task->camera.resolution_y, scale, image,
                                 root.stream),
              "resolve_image");
    if (!out_on_device) hip_check(hipMemcpyAsync(out_rgb, ctx.resolved, bytes, hipMemcpyDeviceToHost, root.stream), "hipMemcpy D2H");
    hip_check(hipStreamSynchronize(root.stream), "hipStreamSynchronize");
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (const Lane& l : lanes) {
            if (!l.active) continue;
            stats->rays += l.stats.rays;
            stats->paths += l.stats.paths;
            stats->searches += l.stats.searches;
            stats->node_visits += l.stats.node_visits;
            stats->triangle_tests += l.stats.triangle_tests;
            stats->wave_node_steps += l.stats.wave_node_steps;
            stats->wave_leaf_steps += l.stats.wave_leaf_steps;
            stats->wave_leaf_rounds += l.stats.wave_leaf_rounds;
            stats->search_kernel_ms += l.stats.search_kernel_ms;
            stats->search_kernel_launches += l.stats.search_kernel_launches;
            stats->bvh_bytes = l.stats.bvh_bytes;
        }
        // the call on the host's clock once its buffers, streams and communicators exist: renders side by side,
        // the reduce, the resolve and the copy back
        stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        stats->reduce_ms = reduce_ms;
    }
}

} // namespace

extern "C" {

size_t pt_last_error(char* buf, size_t cap) {
    const std::string& e = pt::last_error_string();
    if (buf && cap) {
        size_t n = std::min(cap - 1, e.size());
        memcpy(buf, e.data(), n);
        buf[n] = 0;
    }
    return e.size();
}

uint32_t pt_abi_version(void) { return PT_ABI_VERSION; }

PtStatus pt_scene_create(const PtSceneDesc* desc, int device, PtScene** out_scene) {
    return guarded([&] {
        if (!desc || !out_scene) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (!desc->nodes || desc->node_count == 0) throw Error(PT_ERR_SCENE, "scene has no nodes");
        if ((desc->mesh_count && !desc->meshes) || (desc->material_count && !desc->materials) ||
            (desc->image_count && !desc->images) || (desc->tree_count && !desc->trees))
            throw Error(PT_ERR_INVALID_ARGUMENT, "count without array");
        int ndev = 0;
        hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        if (device < 0 || device >= ndev) throw Error(PT_ERR_DEVICE, "no such HIP device");
        DeviceGuard guard(device);
        auto t_create = std::chrono::steady_clock::now();
        auto s = std::make_unique<PtScene>();
        s->device = device;
        memset(&s->info, 0, sizeof(s->info));

        std::vector<ptk::DNode> nodes;
        layout_nodes(*desc, nodes);
        s->node_count = desc->node_count;
        InstanceRun run = find_instance_run(nodes);
        for (uint32_t i = 0; i < desc->node_count; i++) {
            if (nodes[i].type != PT_NODE_MESH) continue;
            if (i >= run.first_node && i < run.end_node) {
                nodes[i].mesh_ordinal = 0xFFFFFFFFu; // reached through the run, never by the walk itself
                continue;
            }
            ptk::MeshTarget t{};
            t.node_index = i;
            t.mesh_id = nodes[i].mesh_id;
            uint32_t above[ptk::kMaxGraphDepth];
            uint32_t n_above = 0;
            for (uint32_t a = nodes[i].parent; a != 0xFFFFFFFFu; a = nodes[a].parent)
                if (nodes[a].type == PT_NODE_TRANSFORM) above[n_above++] = a;
            t.chain_len = n_above;
            for (uint32_t c = 0; c < n_above; c++) t.chain[c] = above[n_above - 1 - c];
            nodes[i].mesh_ordinal = (uint32_t)s->targets.size();
            s->targets.push_back(t);
        }
        for (uint32_t i = 0; i < desc->node_count; i++) {
            nodes[i].only_mesh = 0xFFFFFFFFu;
            if (nodes[i].type != PT_NODE_TRANSFORM) continue;
            uint32_t j = single_mesh_below(nodes, i);
            if (j != 0xFFFFFFFFu) nodes[i].only_mesh = nodes[j].mesh_ordinal;
        }

        // a kd_tree node directly above a mesh node offers its .tree blob to that mesh
        std::vector<const PtTreeBlob*> mesh_blob(desc->mesh_count, nullptr);
        for (uint32_t i = 0; i + 1 < desc->node_count; i++) {
            const PtNodeDesc& nd = desc->nodes[i];
            if (nd.type == PT_NODE_KD_TREE && desc->nodes[i + 1].type == PT_NODE_MESH && nd.tree_id != PT_NO_INDEX) {
                if (nd.tree_id >= desc->tree_count) throw Error(PT_ERR_SCENE, "kd_tree node references a missing tree");
                mesh_blob[desc->nodes[i + 1].mesh_id] = &desc->trees[nd.tree_id];
            }
        }
        std::vector<ptk::DMesh> meshes(desc->mesh_count);
        for (uint32_t i = 0; i < desc->mesh_count; i++) {
            build_mesh(*s, desc->meshes[i], mesh_blob[i], &meshes[i]);
            s->mesh_tri_count.push_back((uint32_t)desc->meshes[i].triangle_count);
        }

        std::vector<const uint8_t*> texels(desc->image_count, nullptr);
        for (uint32_t i = 0; i < desc->image_count; i++) {
            const PtImageDesc& im = desc->images[i];
            if (!im.rgba8 || im.width == 0 || im.height == 0) throw Error(PT_ERR_SCENE, "empty image");
            size_t bytes = (size_t)im.width * im.height * 4;
            void* p = s->dev_alloc(bytes);
            hip_check(hipMemcpy(p, im.rgba8, bytes, hipMemcpyHostToDevice), "hipMemcpy image");
            texels[i] = static_cast<const uint8_t*>(p);
        }
        std::vector<ptk::DMaterial> mats(desc->material_count);
        for (uint32_t i = 0; i < desc->material_count; i++) {
            const PtMaterialDesc& m = desc->materials[i];
            ptk::DMaterial& d = mats[i];
            memset(&d, 0, sizeof(d));
            if (m.type > PT_MATERIAL_PBR) throw Error(PT_ERR_SCENE, "unknown material type");
            d.type = m.type;
            d.color_is_texture = (m.type == PT_MATERIAL_BASE && m.color_is_texture) ? 1u : 0u;
            if (d.color_is_texture) {
                if (m.image_id >= desc->image_count) throw Error(PT_ERR_SCENE, "material references a missing image");
                if (m.uv_mode > PT_UV_REPEAT) throw Error(PT_ERR_SCENE, "unknown uv mode");
                d.texels = texels[m.image_id];
                d.tex_w = desc->images[m.image_id].width;
                d.tex_h = desc->images[m.image_id].height;
                d.uv_mode = m.uv_mode;
            }
            d.color = v3(m.color);
            d.emission = v3(m.emission);
            d.refraction = m.refraction;
            d.reflective = m.reflective;
            d.emissive = m.emissive;
            d.refractive = m.refractive;
            d.albedo = v3(m.albedo);
            d.roughness = m.roughness;
            d.metallic = m.metallic;
            // pbr.rs:92-96
            d.f0 = d.albedo * m.metallic + ptk::splat3(0.04f) * (1.0f - m.metallic);
            d.roughness_sqr = m.roughness * m.roughness;
        }
        s->d.nodes = s->upload(nodes);
        s->d.node_count = desc->node_count;
        s->d.meshes = s->upload(meshes);
        s->d.materials = s->upload(mats);
        s->d.mesh_count = desc->mesh_count;
        s->d.material_count = desc->material_count;
        s->d.targets = s->upload(s->targets);
        s->d.target_count = (uint32_t)s->targets.size();
        memset(&s->d.mesh0, 0, sizeof(s->d.mesh0));
        if (!meshes.empty()) s->d.mesh0 = meshes[0];
        s->d.flat_root = 0;
        if (nodes[0].type == PT_NODE_COLLECTION) {
            bool flat = true;
            for (uint32_t c = 1; c < nodes[0].end && flat; c = nodes[c].end) {
                switch (nodes[c].type) {
                case PT_NODE_SPHERE:
                case PT_NODE_PLANE: break;
                case PT_NODE_TRANSFORM: flat = single_mesh_below(nodes, c) != 0xFFFFFFFFu; break;
                case PT_NODE_MESH:
                case PT_NODE_KD_TREE: {
                    uint32_t j = c;
                    while (j < nodes.size() && nodes[j].type == PT_NODE_KD_TREE) j++;
                    flat = j < nodes.size() && nodes[j].type == PT_NODE_MESH && nodes[c].end == j + 1;
                    break;
                }
                default: flat = false;
                }
            }
            s->d.flat_root = flat ? 1u : 0u;
        }
        memset(&s->d.run, 0, sizeof(s->d.run));
        if (!run.children.empty()) {
            std::vector<ptk::TlasNode> tlas = build_run_tree(run, nodes, meshes);
            s->d.run.tlas = s->upload(tlas);
            s->d.run.children = s->upload(run.children);
            s->d.run.tlas_count = (uint32_t)tlas.size();
            s->d.run.child_count = (uint32_t)run.children.size();
            s->d.run.first_node = run.first_node;
            s->d.run.end_node = run.end_node;
            s->d.run.only_mesh = run.children[0].mesh_id;
            for (const ptk::RunChild& c : run.children)
                if (c.mesh_id != run.children[0].mesh_id) s->d.run.only_mesh = 0xFFFFFFFFu;
            s->run_slots = std::min<uint32_t>((uint32_t)run.children.size(), kRunSlotsPerRay);
        }
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        // set-up for pt_trace_closest_device_static: where this device keeps its copy of the static scratch array
        hip_check(ptk::static_scratch_init(device), "hipGetSymbolAddress(scratch)");
        s->info.mesh_count = desc->mesh_count;
        s->info.bvh_bytes = s->bvh_bytes;
        s->info.instance_run_children = s->d.run.child_count;
        s->info.setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
        *out_scene = s.release();
    });
}

void pt_scene_destroy(PtScene* scene) {
    if (!scene) return;
    try {
        DeviceGuard guard(scene->device);
        delete scene;
    } catch (...) {
    }
}

PtStatus pt_scene_info(const PtScene* scene, PtSceneInfo* out_info) {
    return guarded([&] {
        if (!scene || !out_info) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        *out_info = scene->info;
    });
}

PtStatus pt_device_release_workspace(int device) {
    return guarded([&] {
        int ndev = 0;
        hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        if (device < 0 || device >= ndev) throw Error(PT_ERR_DEVICE, "no such HIP device");
        DeviceGuard guard(device);
        // first what pt_render_multi keeps on this device between calls (a call in progress holds its
        // context's lock and takes the device's below it: the same order here)
        release_multi_buffers(device);
        DeviceState& ds = device_state(device);
        std::lock_guard<std::mutex> lock(ds.mu);
        for (int k = 0; k < kMaxInFlight; k++) {
            wait_idle(ds, k);
            free_workspace(ds.ws[k]);
            free_trace_state(ds.ws[k]);
        }
    });
}

PtStatus pt_render_accumulate_device(PtScene* scene, const PtRenderTask* task, const PtRenderOpts* opts, float* d_accum,
                                     void* stream, PtRenderStats* stats) {
    return guarded([&] {
        if (!scene) throw Error(PT_ERR_INVALID_ARGUMENT, "null scene");
        DeviceGuard guard(scene->device);
        render_accumulate(scene, task, opts, d_accum, static_cast<hipStream_t>(stream), stats);
    });
}

PtStatus pt_resolve_device(const float* d_accum, uint64_t width, uint64_t height, uint64_t total_iterations,
                           float* d_out_rgb, void* stream) {
    return guarded([&] {
        if (!d_accum || !d_out_rgb) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (width == 0 || height == 0 || width > 65536 || height > 65536) throw Error(PT_ERR_INVALID_ARGUMENT, "bad image size");
        // work_group.rs:67-69: 1.0 / (iteration as f32)
        float scale = 1.0f / (float)total_iterations;
        hip_check(ptk::resolve_image(d_accum, (uint32_t)width, (uint32_t)height, scale, d_out_rgb,
                                     static_cast<hipStream_t>(stream)),
                  "resolve_image");
    });
}

PtStatus pt_render(PtScene* scene, const PtRenderTask* task, const PtRenderOpts* opts, float* out_rgb,
                   PtRenderStats* stats) {
    return guarded([&] {
        if (!scene || !task || !out_rgb) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        DeviceGuard guard(scene->device);
        (void)make_camera(task->camera);
        size_t floats = (size_t)task->camera.resolution_x * task->camera.resolution_y * 3;
        DevBuf accum(floats * sizeof(float)), out(floats * sizeof(float));
        hip_check(hipMemsetAsync(accum.p, 0, floats * sizeof(float), nullptr), "hipMemsetAsync");
        PtRenderStats local;
        memset(&local, 0, sizeof(local));
        render_accumulate(scene, task, opts, static_cast<float*>(accum.p), nullptr, stats ? stats : &local);
        float scale = 1.0f / (float)task->iterations;
        hip_check(ptk::resolve_image(static_cast<float*>(accum.p), (uint32_t)task->camera.resolution_x,
                                     (uint32_t)task->camera.resolution_y, scale, static_cast<float*>(out.p), nullptr),
                  "resolve_image");
        hip_check(hipMemcpy(out_rgb, out.p, floats * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    });
}

PtStatus pt_shard(const PtRenderTask* task, const PtRenderOpts* opts, uint32_t mode, uint32_t world, uint32_t rank,
                  PtRenderTask* out_task, PtRenderOpts* out_opts, uint32_t* out_active) {
    return guarded([&] {
        if (!task || !out_task || !out_opts || !out_active) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        PtRenderOpts base;
        memset(&base, 0, sizeof(base));
        if (opts) base = *opts;
        *out_active = shard(*task, base, mode, world, rank, out_task, out_opts) ? 1u : 0u;
    });
}

PtStatus pt_render_multi(PtScene** scenes, int n_devices, const PtRenderTask* task, const PtRenderOpts* opts,
                         uint32_t mode, float* out_rgb, PtRenderStats* stats) {
    return guarded([&] { render_multi(scenes, n_devices, task, opts, mode, out_rgb, stats); });
}

PtStatus pt_discovery_request_encode(uint8_t* out, size_t cap, size_t* out_len) {
    return guarded([&] {
        (void)out;
        (void)cap;
        if (!out_len) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        *out_len = 0; // `struct Request {}` (lib.rs:84-85) has no fields: postcard writes nothing
    });
}

PtStatus pt_discovery_request_decode(const uint8_t* bytes, size_t len) {
    (void)bytes;
    (void)len;
    return PT_OK; // nothing to read, trailing bytes are ignored (main.rs:104 hands postcard the whole datagram)
}

PtStatus pt_discovery_response_encode(uint16_t websocket_port, uint8_t* out, size_t cap, size_t* out_len) {
    return guarded([&] {
        if (!out || !out_len) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        std::vector<uint8_t> bytes = pt::discovery_response_encode(websocket_port);
        if (cap < bytes.size()) throw Error(PT_ERR_INVALID_ARGUMENT, "output buffer too small (three bytes always suffice)");
        memcpy(out, bytes.data(), bytes.size());
        *out_len = bytes.size();
    });
}

PtStatus pt_discovery_response_decode(const uint8_t* bytes, size_t len, uint16_t* out_websocket_port) {
    return guarded([&] {
        if ((len && !bytes) || !out_websocket_port) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (!pt::discovery_response_decode(bytes, len, out_websocket_port))
            throw Error(PT_ERR_PARSE, "not a postcard-encoded discovery::Response (bad or truncated varint)");
    });
}

PtStatus pt_scene_md5(const char* root_dir, const char* scene_path, char* out_hex33) {
    return guarded([&] {
        if (!root_dir || !scene_path || !out_hex33) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        std::string digest = pt::scene_md5(pt::FileStore(root_dir), scene_path);
        memcpy(out_hex33, digest.c_str(), 33);
    });
}

PtStatus pt_worker_create(const char* store_root, int device, PtWorker** out_worker) {
    return guarded([&] {
        if (!store_root || !out_worker) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        auto w = std::make_unique<PtWorker>();
        w->store_root = store_root;
        w->device = device;
        *out_worker = w.release();
    });
}

void pt_worker_destroy(PtWorker* worker) { delete worker; }

uint32_t pt_worker_cached_scene_count(const PtWorker* worker) { return worker ? (uint32_t)worker->cached_scenes.size() : 0u; }

PtStatus pt_worker_handle_message(PtWorker* worker, const char* task_json, const PtRenderOpts* opts, uint8_t** out_bytes,
                                  uint64_t* out_size) {
    PtStatus nested = PT_OK; // a failure of an entry point called from here: its message is already in place
    PtStatus st = guarded([&] {
        if (!worker || !task_json || !out_bytes || !out_size) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        std::lock_guard<std::mutex> lock(worker->mu);
        // connection_loop, main.rs:68-93: a text frame holding the RenderTask ...
        PtRenderTask task;
        std::string scene_path, scene_md5;
        try {
            pt::parse_render_task(task_json, &task, &scene_path, &scene_md5);
        } catch (const std::exception& e) {
            throw Error(PT_ERR_PARSE, std::string("Failed to decode render task: ") + e.what());
        }
        // the wire form is RenderTask (render_task.rs:23-29): serde refuses a frame without scene_md5, main.rs:79
        if (scene_md5.empty()) throw Error(PT_ERR_PARSE, "Failed to decode render task: missing field `scene_md5`");
        for (char c : scene_md5)
            if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F')))
                throw Error(PT_ERR_INVALID_ARGUMENT, "scene_md5 is not a hex digest");
        // ... Worker::render, lib.rs:31-48: the scene from the cache, or loaded from its bucket
        const std::string key = scene_md5;
        auto it = worker->cached_scenes.find(key);
        if (it == worker->cached_scenes.end()) {
            // the names in the frame and in the files it leads to are a peer's: they stay inside the bucket
            pt::FileStore store(worker->store_root + "/" + scene_md5, /*confined=*/true);
            PtWorker::Entry entry;
            entry.host = pt::load_scene(store, scene_path);
            try {
                (void)cache_trees(*entry.host, store);
            } catch (const Error& e) {
                if (e.status != PT_ERR_IO) throw; // a read-only bucket is no reason to refuse the task
            }
            nested = pt_scene_create(&entry.host->desc, worker->device, &entry.scene);
            if (nested != PT_OK) throw Error(nested, pt::last_error_string());
            it = worker->cached_scenes.emplace(key, std::move(entry)).first;
        }
        const size_t floats = (size_t)task.camera.resolution_x * task.camera.resolution_y * 3;
        if (task.camera.resolution_x > 0xFFFFFFFFull || task.camera.resolution_y > 0xFFFFFFFFull)
            throw Error(PT_ERR_UNSUPPORTED, "resolution does not fit the result's u32 header");
        // ... the image as the binary frame of RenderedImage::to_bytes, lib.rs:56-62
        uint8_t* bytes = static_cast<uint8_t*>(malloc(8 + floats * sizeof(float) + 16));
        if (!bytes) throw std::bad_alloc();
        struct Guard {
            uint8_t* p;
            ~Guard() { free(p); }
        } guard{bytes};
        nested = pt_render(it->second.scene, &task, opts, reinterpret_cast<float*>(bytes + 8), nullptr);
        if (nested != PT_OK) throw Error(nested, pt::last_error_string());
        const uint32_t w = (uint32_t)task.camera.resolution_x, h = (uint32_t)task.camera.resolution_y;
        memcpy(bytes, &w, 4); // little-endian target: the bytes are already in wire order
        memcpy(bytes + 4, &h, 4);
        *out_bytes = bytes;
        *out_size = 8 + floats * sizeof(float);
        guard.p = nullptr;
    });
    return st;
}

struct PtServer {
    std::unique_ptr<pt::Server> server;
};

PtStatus pt_worker_serve_start(PtWorker* worker, const char* bind_addr, uint16_t websocket_port, uint16_t udp_port,
                               PtServer** out_server) {
    return guarded([&] {
        if (!worker || !out_server) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        // one round of connection_loop: the text frame through the worker, its bytes as the reply
        pt::MessageHandler handler = [worker](const std::string& text, std::vector<uint8_t>* reply, std::string* error) {
            uint8_t* bytes = nullptr;
            uint64_t size = 0;
            if (pt_worker_handle_message(worker, text.c_str(), nullptr, &bytes, &size) != PT_OK) {
                *error = pt::last_error_string();
                return false;
            }
            reply->assign(bytes, bytes + size);
            pt_buffer_free(bytes);
            return true;
        };
        auto s = std::make_unique<PtServer>();
        s->server = std::make_unique<pt::Server>(bind_addr && *bind_addr ? bind_addr : "0.0.0.0", websocket_port, udp_port,
                                                 std::move(handler));
        *out_server = s.release();
    });
}

PtStatus pt_server_ports(const PtServer* server, uint16_t* out_websocket_port, uint16_t* out_udp_port) {
    return guarded([&] {
        if (!server) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (out_websocket_port) *out_websocket_port = server->server->websocket_port();
        if (out_udp_port) *out_udp_port = server->server->udp_port();
    });
}

PtStatus pt_server_stats(const PtServer* server, PtServerStats* out_stats, char* last_error, size_t last_error_cap) {
    return guarded([&] {
        if (!server || !out_stats) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        const pt::ServerCounters c = server->server->counters();
        out_stats->connections = c.connections;
        out_stats->handshakes = c.handshakes;
        out_stats->messages_answered = c.messages_answered;
        out_stats->messages_failed = c.messages_failed;
        out_stats->datagrams_answered = c.datagrams_answered;
        if (last_error && last_error_cap) {
            const std::string e = server->server->last_error();
            const size_t n = std::min(e.size(), last_error_cap - 1);
            memcpy(last_error, e.data(), n);
            last_error[n] = 0;
        }
    });
}

void pt_server_stop(PtServer* server) {
    if (!server) return;
    try {
        server->server->stop();
    } catch (...) {
    }
    delete server;
}

PtStatus pt_worker_serve(PtWorker* worker, const char* bind_addr, uint16_t websocket_port, uint16_t udp_port,
                         const int32_t* stop_flag) {
    PtServer* server = nullptr;
    PtStatus st = pt_worker_serve_start(worker, bind_addr, websocket_port, udp_port, &server);
    if (st != PT_OK) return st;
    while (!stop_flag || *static_cast<const volatile int32_t*>(stop_flag) == 0) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    pt_server_stop(server);
    return PT_OK;
}

PtStatus pt_frame_create(uint32_t width, uint32_t height, int device, PtFrame** out_frame) {
    return guarded([&] {
        if (!out_frame) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (width == 0 || height == 0 || width > 65536 || height > 65536) throw Error(PT_ERR_INVALID_ARGUMENT, "bad image size");
        int ndev = 0;
        hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
        if (device < 0 || device >= ndev) throw Error(PT_ERR_DEVICE, "no such HIP device");
        DeviceGuard guard(device);
        auto f = std::make_unique<PtFrame>();
        f->device = device;
        f->width = width;
        f->height = height;
        size_t bytes = f->floats() * sizeof(float);
        hip_check(hipMalloc(reinterpret_cast<void**>(&f->sum), bytes), "hipMalloc frame");
        hip_check(hipMalloc(reinterpret_cast<void**>(&f->staging), bytes), "hipMalloc frame");
        hip_check(hipMalloc(reinterpret_cast<void**>(&f->rgba), (size_t)width * height * 4), "hipMalloc frame");
        hip_check(hipMemset(f->sum, 0, bytes), "hipMemset"); // Rgb32FImage::new is zeroed, frame.rs:27
        *out_frame = f.release();
    });
}

void pt_frame_destroy(PtFrame* frame) {
    if (!frame) return;
    (void)guarded([&] {
        DeviceGuard guard(frame->device);
        delete frame;
    });
}

uint64_t pt_frame_render_count(const PtFrame* frame) { return frame ? frame->count : 0; }

PtStatus pt_frame_add_render(PtFrame* frame, const float* rgb) {
    return guarded([&] {
        if (!frame || !rgb) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        DeviceGuard guard(frame->device);
        hip_check(hipMemcpy(frame->staging, rgb, frame->floats() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy H2D");
        hip_check(ptk::frame_add(frame->sum, frame->staging, frame->floats(), nullptr), "frame_add");
        hip_check(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
        frame->count += 1;
    });
}

PtStatus pt_frame_add_render_device(PtFrame* frame, const float* d_rgb, void* stream) {
    return guarded([&] {
        if (!frame || !d_rgb) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        DeviceGuard guard(frame->device);
        hip_check(ptk::frame_add(frame->sum, d_rgb, frame->floats(), static_cast<hipStream_t>(stream)), "frame_add");
        frame->count += 1;
    });
}

PtStatus pt_frame_image(PtFrame* frame, float* out_rgb, uint8_t* out_rgba) {
    return guarded([&] {
        if (!frame) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (!out_rgb && !out_rgba) return;
        DeviceGuard guard(frame->device);
        // frame.rs:74: render_count = count.max(1), then `ch / render_count as f32`
        float divisor = (float)std::max<uint64_t>(frame->count, 1);
        size_t pixels = (size_t)frame->width * frame->height;
        // the default stream waits for work other streams queued through pt_frame_add_render_device
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        hip_check(ptk::frame_image(frame->sum, pixels, divisor, out_rgb ? frame->staging : nullptr,
                                   out_rgba ? frame->rgba : nullptr, nullptr),
                  "frame_image");
        if (out_rgb) hip_check(hipMemcpy(out_rgb, frame->staging, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        if (out_rgba) hip_check(hipMemcpy(out_rgba, frame->rgba, pixels * 4, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    });
}

PtStatus pt_frame_image_device(PtFrame* frame, float* d_out_rgb, uint8_t* d_out_rgba, void* stream) {
    return guarded([&] {
        if (!frame) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (reinterpret_cast<uintptr_t>(d_out_rgba) & 3u) throw Error(PT_ERR_INVALID_ARGUMENT, "d_out_rgba is not 4-byte aligned");
        DeviceGuard guard(frame->device);
        float divisor = (float)std::max<uint64_t>(frame->count, 1);
        hip_check(ptk::frame_image(frame->sum, (size_t)frame->width * frame->height, divisor, d_out_rgb, d_out_rgba,
                                   static_cast<hipStream_t>(stream)),
                  "frame_image");
    });
}

PtStatus pt_trace_closest(PtScene* scene, const PtRay* rays, size_t n, uint32_t flags, PtHit* out_hits) {
    return guarded([&] {
        if (!scene || (n && (!rays || !out_hits))) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (n == 0) return;
        const bool brute = (flags & PT_FLAG_BRUTE_FORCE) != 0;
        const bool wavefront = (flags & (PT_FLAG_WAVEFRONT | PT_FLAG_WAVEFRONT_SPLIT)) != 0;
        if (wavefront) {
            if (brute) throw Error(PT_ERR_INVALID_ARGUMENT, "the wavefront route has no linear-scan mode");
            if (n > 0x7FFFFFFFull) throw Error(PT_ERR_UNSUPPORTED, "more than 2^31 rays in one wavefront trace");
            // the queues of a render carry no ray.max: every ray of a render has f32::MAX (work_group.rs:43, camera.rs:82)
            for (size_t i = 0; i < n; i++)
                if (!(rays[i].max == 3.4028234663852886e38f))
                    throw Error(PT_ERR_UNSUPPORTED, "the wavefront route takes rays with max = f32::MAX only");
        }
        DeviceGuard guard(scene->device);
        DeviceState& ds = device_state(scene->device);
        std::lock_guard<std::mutex> lock(ds.mu);
        DevBuf d_rays(n * sizeof(PtRay)), d_hits(n * sizeof(PtHit));
        hip_check(hipMemcpy(d_rays.p, rays, n * sizeof(PtRay), hipMemcpyHostToDevice), "hipMemcpy H2D");
        ensure_trace_state(ds, 0);
        if (wavefront) {
            const uint32_t waves = ds.ws[0].persistent_blocks * (ptk::kTraceBlock / 64);
            const uint64_t records = (uint64_t)ptk::injected_region_capacity(n, waves) * waves;
            if (!ensure_workspace(ds, 0, records, n, scene->result_arrays(), scene->run_slots))
                throw Error(PT_ERR_DEVICE, "out of device memory for the trace's queues");
        }
        stream_acquire(ds, 0, nullptr);
        if (wavefront)
            hip_check(ptk::trace_closest_wavefront(scene->d, scene->targets, scene->run_slots, static_cast<const PtRay*>(d_rays.p),
                                                   n, static_cast<PtHit*>(d_hits.p), (flags & PT_FLAG_WAVEFRONT_SPLIT) != 0,
                                                   ds.ws[0], nullptr),
                      "trace_closest_wavefront");
        else
            hip_check(ptk::trace_closest(scene->d, static_cast<const PtRay*>(d_rays.p), n, brute, static_cast<PtHit*>(d_hits.p),
                                         ds.ws[0], nullptr),
                      "trace_closest");
        stream_release(ds, 0, nullptr);
        hip_check(hipMemcpy(out_hits, d_hits.p, n * sizeof(PtHit), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    });
}

PtStatus pt_trace_closest_device_static(PtScene* scene, const PtRay* d_rays, size_t n, uint32_t flags, PtHit* d_hits,
                                        void* stream) {
    return guarded([&] {
        if (!scene || (n && (!d_rays || !d_hits))) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (flags & ~PT_FLAG_BRUTE_FORCE) throw Error(PT_ERR_INVALID_ARGUMENT, "pt_trace_closest_device_static takes PT_FLAG_BRUTE_FORCE only");
        if (n > PT_STATIC_TRACE_MAX_RAYS)
            throw Error(PT_ERR_INVALID_ARGUMENT, "pt_trace_closest_device_static: more than PT_STATIC_TRACE_MAX_RAYS (16777216) rays in one call");
        if (n == 0) return;
        DeviceGuard guard(scene->device);
        hip_check(ptk::trace_closest_static(scene->device, scene->d, d_rays, n, (flags & PT_FLAG_BRUTE_FORCE) != 0, d_hits,
                                            static_cast<hipStream_t>(stream)),
                  "trace_closest_static");
    });
}

PtStatus pt_generate_rays(const PtCamera* camera, uint64_t seed, const uint32_t* pixel_xy, const uint64_t* sample_index,
                          size_t n, int device, PtRay* out_rays) {
    return guarded([&] {
        if (!camera || (n && (!pixel_xy || !sample_index || !out_rays))) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (n == 0) return;
        ptk::DCamera cam = make_camera(*camera);
        for (size_t i = 0; i < n; i++)
            if (pixel_xy[2 * i] >= cam.width || pixel_xy[2 * i + 1] >= cam.height)
                throw Error(PT_ERR_INVALID_ARGUMENT, "pixel outside the frame");
        DeviceGuard guard(device);
        DevBuf d_px(n * 8), d_s(n * 8), d_r(n * sizeof(PtRay));
        hip_check(hipMemcpy(d_px.p, pixel_xy, n * 8, hipMemcpyHostToDevice), "hipMemcpy H2D");
        hip_check(hipMemcpy(d_s.p, sample_index, n * 8, hipMemcpyHostToDevice), "hipMemcpy H2D");
        hip_check(ptk::generate_rays(cam, seed, static_cast<const uint32_t*>(d_px.p), static_cast<const uint64_t*>(d_s.p), n,
                                     static_cast<PtRay*>(d_r.p), nullptr),
                  "generate_rays");
        hip_check(hipMemcpy(out_rays, d_r.p, n * sizeof(PtRay), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    });
}

PtStatus pt_shade(PtScene* scene, const PtVec3* directions, const PtHit* hits, const uint64_t* stream_id, uint64_t seed,
                  size_t n, uint32_t* out_kind, PtVec3* out_a, PtVec3* out_direction) {
    return guarded([&] {
        if (!scene || (n && (!directions || !hits || !stream_id || !out_kind || !out_a || !out_direction)))
            throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (n == 0) return;
        for (size_t i = 0; i < n; i++)
            if (hits[i].material_id >= scene->d.material_count) throw Error(PT_ERR_INVALID_ARGUMENT, "material id out of range");
        DeviceGuard guard(scene->device);
        DevBuf d_dir(n * sizeof(PtVec3)), d_hit(n * sizeof(PtHit)), d_sid(n * 8), d_kind(n * 4), d_a(n * sizeof(PtVec3)),
            d_o(n * sizeof(PtVec3));
        hip_check(hipMemcpy(d_dir.p, directions, n * sizeof(PtVec3), hipMemcpyHostToDevice), "hipMemcpy H2D");
        hip_check(hipMemcpy(d_hit.p, hits, n * sizeof(PtHit), hipMemcpyHostToDevice), "hipMemcpy H2D");
        hip_check(hipMemcpy(d_sid.p, stream_id, n * 8, hipMemcpyHostToDevice), "hipMemcpy H2D");
        hip_check(ptk::shade(scene->d, static_cast<const PtVec3*>(d_dir.p), static_cast<const PtHit*>(d_hit.p),
                             static_cast<const uint64_t*>(d_sid.p), seed, n, static_cast<uint32_t*>(d_kind.p),
                             static_cast<PtVec3*>(d_a.p), static_cast<PtVec3*>(d_o.p), nullptr),
                  "shade");
        hip_check(hipMemcpy(out_kind, d_kind.p, n * 4, hipMemcpyDeviceToHost), "hipMemcpy D2H");
        hip_check(hipMemcpy(out_a, d_a.p, n * sizeof(PtVec3), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        hip_check(hipMemcpy(out_direction, d_o.p, n * sizeof(PtVec3), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    });
}

PtStatus pt_host_scene_load(const char* root_dir, const char* scene_path, PtHostScene** out) {
    return guarded([&] {
        if (!root_dir || !scene_path || !out) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        pt::FileStore store(root_dir);
        auto hs = std::make_unique<PtHostScene>();
        hs->scene = pt::load_scene(store, scene_path);
        *out = hs.release();
    });
}

PtStatus pt_host_scene_cache_trees(PtHostScene* scene, const char* root_dir, uint32_t* out_written) {
    return guarded([&] {
        if (!scene || !scene->scene || !root_dir) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        uint32_t written = cache_trees(*scene->scene, pt::FileStore(root_dir));
        if (out_written) *out_written = written;
    });
}

const PtSceneDesc* pt_host_scene_desc(const PtHostScene* scene) { return scene ? &scene->scene->desc : nullptr; }

void pt_host_scene_destroy(PtHostScene* scene) { delete scene; }

PtStatus pt_render_task_parse(const char* json_text, PtRenderTask* out_task, char* scene_path_buf, size_t scene_path_cap) {
    return guarded([&] {
        if (!json_text || !out_task) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        std::string path;
        pt::parse_render_task(json_text, out_task, &path);
        if (scene_path_buf && scene_path_cap) {
            if (path.size() + 1 > scene_path_cap) throw Error(PT_ERR_INVALID_ARGUMENT, "scene path buffer too small");
            memcpy(scene_path_buf, path.c_str(), path.size() + 1);
        }
    });
}

PtStatus pt_tree_build(const PtMeshDesc* mesh, uint8_t** out_data, uint64_t* out_size) {
    return guarded([&] {
        if (!mesh || !out_data || !out_size || (mesh->triangle_count && !mesh->positions))
            throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        pt::BuiltBvh bvh = pt::build_bvh(mesh->positions, mesh->triangle_count, pt::default_bvh_tuning().max_leaf);
        std::vector<uint8_t> blob = pt::serialize_tree(bvh, mesh->positions, mesh->triangle_count);
        uint8_t* p = static_cast<uint8_t*>(malloc(blob.size() ? blob.size() : 1));
        if (!p) throw std::bad_alloc();
        memcpy(p, blob.data(), blob.size());
        *out_data = p;
        *out_size = blob.size();
    });
}

void pt_buffer_free(uint8_t* data) { free(data); }

PtStatus pt_tree_validate(const PtMeshDesc* mesh, const uint8_t* data, uint64_t size) {
    return guarded([&] {
        if (!mesh || !data || (mesh->triangle_count && !mesh->positions)) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        pt::BuiltBvh bvh;
        if (!pt::deserialize_tree(data, size, mesh->positions, mesh->triangle_count, &bvh))
            throw Error(PT_ERR_PARSE, "not a valid tree for this mesh");
    });
}

PtStatus pt_rendered_image_from_bytes(const uint8_t* bytes, size_t size, uint32_t* out_width, uint32_t* out_height,
                                      float* out_rgb, size_t out_cap_floats) {
    return guarded([&] {
        if (!bytes || !out_width || !out_height) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        if (size < 8) throw Error(PT_ERR_PARSE, "rendered image shorter than its header");
        auto le32 = [&](size_t at) {
            return (uint32_t)bytes[at] | ((uint32_t)bytes[at + 1] << 8) | ((uint32_t)bytes[at + 2] << 16) | ((uint32_t)bytes[at + 3] << 24);
        };
        uint32_t w = le32(0), h = le32(4);
        size_t floats = (size_t)w * h * 3;
        if ((size - 8) / 4 < floats) throw Error(PT_ERR_PARSE, "rendered image holds fewer floats than its header announces");
        *out_width = w;
        *out_height = h;
        if (!out_rgb) return;
        if (out_cap_floats < floats) throw Error(PT_ERR_INVALID_ARGUMENT, "output buffer too small");
        for (size_t i = 0; i < floats; i++) {
            uint32_t u = le32(8 + 4 * i);
            memcpy(out_rgb + i, &u, 4);
        }
    });
}

PtStatus pt_rendered_image_to_bytes(const float* rgb, uint32_t width, uint32_t height, uint8_t* out, size_t out_cap) {
    return guarded([&] {
        if (!rgb || !out) throw Error(PT_ERR_INVALID_ARGUMENT, "null argument");
        size_t floats = (size_t)width * height * 3;
        if (out_cap < 8 + floats * 4) throw Error(PT_ERR_INVALID_ARGUMENT, "output buffer too small");
        // lib.rs:56-62; this target is little-endian, so the f32 bytes are copied as they are
        memcpy(out, &width, 4);
        memcpy(out + 4, &height, 4);
        memcpy(out + 8, rgb, floats * 4);
    });
}

} // extern "C"
