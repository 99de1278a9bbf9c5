// Host-side scene model: what Scene::load (worker/src/scene/mod.rs:58-112)
// produces, held in plain arrays that a PtSceneDesc can point into.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/pathtracing_amd.h"

namespace pt {

struct Error : std::runtime_error {
    PtStatus status;
    Error(PtStatus s, const std::string& what) : std::runtime_error(what), status(s) {}
};

void set_last_error(const std::string& message);

// resource::Mesh before TriangleUninit::init: one entry per triangle, OBJ face order.
struct HostMesh {
    uint64_t triangle_count = 0;
    std::vector<float> positions;      // 9 per triangle
    std::vector<float> normals;        // 9 per triangle when any triangle has normals, else empty
    std::vector<uint8_t> normal_valid; // 1 per triangle when `normals` is non-empty
    std::vector<float> uvs;            // 6 per triangle when any triangle has uvs, else empty
};

struct HostImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba8;
};

// file_store.rs:27-35 (`fetch_file`) against a local directory.
class FileStore {
public:
    // `confined`: names come from an untrusted peer (a task frame, and the files it names); a name that is
    // absolute or has a ".." component is refused (PT_ERR_INVALID_ARGUMENT) instead of leaving the directory.
    explicit FileStore(std::string root_dir, bool confined = false) : root_(std::move(root_dir)), confined_(confined) {}
    std::vector<uint8_t> fetch_file(const std::string& path) const;
    bool exists(const std::string& path) const;
    // Replaces (or creates) the file: written beside it under a temporary name, then renamed over it.
    void write_file(const std::string& path, const std::vector<uint8_t>& data) const;

private:
    std::string full_path(const std::string& path) const;
    std::string root_;
    bool confined_ = false;
};

struct HostScene {
    std::vector<PtNodeDesc> nodes;
    std::vector<HostMesh> meshes;
    std::vector<PtMaterialDesc> materials;
    std::vector<HostImage> images;
    std::vector<std::vector<uint8_t>> trees;
    // names in id order, for diagnostics
    std::vector<std::string> mesh_paths, material_paths, image_paths, tree_paths;

    // views handed across the C ABI; valid after finalize()
    std::vector<PtMeshDesc> mesh_descs;
    std::vector<PtImageDesc> image_descs;
    std::vector<PtTreeBlob> tree_blobs;
    PtSceneDesc desc{};
    void finalize();
};

// obj_loader.rs:9-82 with tobj 3.2 semantics (triangulate, separate index streams).
HostMesh load_obj(const uint8_t* data, size_t size);

// image.rs:13-19: decode to RGBA8.  PNG only (the only format the shipped scenes use).
HostImage decode_png(const uint8_t* data, size_t size);

// Scene::load.
std::unique_ptr<HostScene> load_scene(const FileStore& store, const std::string& scene_path);

// api/render_task.rs:5-29.
// `scene_md5` (may be null) receives the wire form's field of that name, or "" for the file form
// (RenderTaskUninit, render_task.rs:5-10).
void parse_render_task(const std::string& json_text, PtRenderTask* out, std::string* scene_path,
                       std::string* scene_md5 = nullptr);

} // namespace pt
