// Scene::load (worker/src/scene/mod.rs:58-112) and the RenderTask schema
// (worker/src/api/render_task.rs:5-49, worker/src/camera.rs:8-48), reading the
// same files from a local directory instead of MongoDB GridFS
// (worker/src/file_store.rs:27-35).
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "host_scene.hpp"
#include "json.hpp"

namespace pt {

namespace {
thread_local std::string g_last_error;
}
void set_last_error(const std::string& message) { g_last_error = message; }
const std::string& last_error_string() { return g_last_error; }

// ----------------------------------------------------------------- FileStore

static std::string join_path(const std::string& root, const std::string& path) {
    if (root.empty()) return path;
    if (root.back() == '/') return root + path;
    return root + "/" + path;
}

std::string FileStore::full_path(const std::string& path) const {
    if (confined_) {
        bool escapes = path.empty() || path[0] == '/' || path[0] == '\\' || path.find('\0') != std::string::npos;
        for (size_t i = 0; i < path.size() && !escapes;) { // component by component
            size_t j = path.find_first_of("/\\", i);
            if (j == std::string::npos) j = path.size();
            if (path.compare(i, j - i, "..") == 0) escapes = true;
            i = j + 1;
        }
        if (escapes) throw Error(PT_ERR_INVALID_ARGUMENT, "file name '" + path + "' leaves the scene's bucket");
    }
    return join_path(root_, path);
}

bool FileStore::exists(const std::string& path) const {
    FILE* f = fopen(full_path(path).c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

std::vector<uint8_t> FileStore::fetch_file(const std::string& path) const {
    std::string full = full_path(path);
    FILE* f = fopen(full.c_str(), "rb");
    if (!f) throw Error(PT_ERR_IO, "cannot open '" + full + "'");
    std::vector<uint8_t> out;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
    bool bad = ferror(f) != 0;
    fclose(f);
    if (bad) throw Error(PT_ERR_IO, "read error on '" + full + "'");
    return out;
}

void FileStore::write_file(const std::string& path, const std::vector<uint8_t>& data) const {
    std::string full = full_path(path);
    std::string tmp = full + ".tmp" + std::to_string((long long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) throw Error(PT_ERR_IO, "cannot create '" + tmp + "'");
    size_t n = data.empty() ? 0 : fwrite(data.data(), 1, data.size(), f);
    bool bad = n != data.size() || fflush(f) != 0;
    bad = (fclose(f) != 0) || bad;
    if (bad || rename(tmp.c_str(), full.c_str()) != 0) {
        remove(tmp.c_str());
        throw Error(PT_ERR_IO, "cannot write '" + full + "'");
    }
}

// --------------------------------------------------------------- JSON helpers

namespace {

using ptjson::Value;

[[noreturn]] void bad(const std::string& what) { throw Error(PT_ERR_PARSE, what); }

const Value& field(const Value& obj, const char* key, const char* ctx) {
    if (obj.kind != Value::Object) bad(std::string(ctx) + ": expected an object");
    const Value* v = obj.find(key);
    if (!v) bad(std::string(ctx) + ": missing field `" + key + "`");
    return *v;
}

// serde_json hands f32 fields the f64 value cast with `as f32`
float as_f32(const Value& v, const char* ctx) {
    if (v.kind != Value::Number) bad(std::string(ctx) + ": expected a number");
    return (float)v.num;
}

uint64_t as_usize(const Value& v, const char* ctx) {
    if (v.kind != Value::Number || !v.is_integer || v.negative) bad(std::string(ctx) + ": expected a non-negative integer");
    return v.uint_value;
}

const std::string& as_string(const Value& v, const char* ctx) {
    if (v.kind != Value::String) bad(std::string(ctx) + ": expected a string");
    return v.str;
}

// math Vec3 deserialises from a 3-element array (math/src/lib.rs:214-220)
PtVec3 as_vec3(const Value& v, const char* ctx) {
    if (v.kind != Value::Array || v.arr.size() != 3) bad(std::string(ctx) + ": expecting [<x>, <y>, <z>] array");
    return {as_f32(v.arr[0], ctx), as_f32(v.arr[1], ctx), as_f32(v.arr[2], ctx)};
}

PtMat3 as_mat3(const Value& v, const char* ctx) {
    if (v.kind != Value::Array || v.arr.size() != 3) bad(std::string(ctx) + ": expecting [<row0>, <row1>, <row2>] array");
    return {as_vec3(v.arr[0], ctx), as_vec3(v.arr[1], ctx), as_vec3(v.arr[2], ctx)};
}

PtMat4 as_mat4(const Value& v, const char* ctx) {
    if (v.kind != Value::Array || v.arr.size() != 4)
        bad(std::string(ctx) + ": expecting [<row0>, <row1>, <row2>, <row3>] array");
    PtMat4 m;
    for (int r = 0; r < 4; r++) {
        const Value& row = v.arr[r];
        if (row.kind != Value::Array || row.arr.size() != 4) bad(std::string(ctx) + ": expecting [<x>, <y>, <z>, <w>] array");
        for (int c = 0; c < 4; c++) m.m[r][c] = as_f32(row.arr[c], ctx);
    }
    return m;
}

// ReferenceMapping (scene/resource/mod.rs:20-97): dense ids per resource type
// in first-touch order.
struct ReferenceCollection {
    std::map<std::string, uint32_t> ids;
    std::vector<std::string> paths; // id -> path
    uint32_t get_id_or_insert(const std::string& path) {
        auto it = ids.find(path);
        if (it != ids.end()) return it->second;
        uint32_t id = (uint32_t)paths.size();
        ids.emplace(path, id);
        paths.push_back(path);
        return id;
    }
};

struct Loader {
    const FileStore& store;
    HostScene& scene;
    ReferenceCollection meshes, materials, trees, images;

    // SceneNodeUnloaded::init, depth first (hierarchy/*.rs `init`)
    void add_node(const Value& v, int depth) {
        if (depth > 64) throw Error(PT_ERR_SCENE, "scene graph too deep");
        const std::string& type = as_string(field(v, "type", "scene node"), "scene node type");
        PtNodeDesc nd;
        memset(&nd, 0, sizeof(nd));
        nd.tree_id = PT_NO_INDEX;
        size_t me = scene.nodes.size();
        if (type == "node_collection") { // node_collection.rs:30-38
            const Value& children = field(v, "children", "node_collection");
            if (children.kind != Value::Array) bad("node_collection: `children` must be an array");
            nd.type = PT_NODE_COLLECTION;
            nd.child_count = (uint32_t)children.arr.size();
            scene.nodes.push_back(nd);
            for (const Value& c : children.arr) add_node(c, depth + 1);
        } else if (type == "transform") { // transform.rs:33-40
            nd.type = PT_NODE_TRANSFORM;
            nd.child_count = 1;
            nd.matrix = as_mat4(field(v, "matrix", "transform"), "transform.matrix");
            const Value& child = field(v, "child", "transform");
            scene.nodes.push_back(nd);
            add_node(child, depth + 1);
        } else if (type == "kd_tree") { // kd_tree.rs:32-42: tree path first, then the child
            nd.type = PT_NODE_KD_TREE;
            nd.child_count = 1;
            nd.tree_id = trees.get_id_or_insert(as_string(field(v, "path", "kd_tree"), "kd_tree.path"));
            const Value& child = field(v, "child", "kd_tree");
            scene.nodes.push_back(nd);
            add_node(child, depth + 1);
        } else if (type == "mesh") { // hierarchy/mesh.rs:41-56: material first, then the mesh
            nd.type = PT_NODE_MESH;
            const std::string& path = as_string(field(v, "path", "mesh"), "mesh.path");
            const std::string& mat = as_string(field(v, "material", "mesh"), "mesh.material");
            nd.material_id = materials.get_id_or_insert(mat);
            nd.mesh_id = meshes.get_id_or_insert(path);
            scene.nodes.push_back(nd);
        } else if (type == "sphere") { // sphere.rs:42-54
            nd.type = PT_NODE_SPHERE;
            nd.center = as_vec3(field(v, "center", "sphere"), "sphere.center");
            nd.radius = as_f32(field(v, "radius", "sphere"), "sphere.radius");
            nd.material_id = materials.get_id_or_insert(as_string(field(v, "material", "sphere"), "sphere.material"));
            scene.nodes.push_back(nd);
        } else if (type == "plane") { // plane.rs:41-67
            nd.type = PT_NODE_PLANE;
            nd.point = as_vec3(field(v, "point", "plane"), "plane.point");
            nd.normal = as_vec3(field(v, "normal", "plane"), "plane.normal");
            nd.tangent = as_vec3(field(v, "tangent", "plane"), "plane.tangent");
            nd.bitangent = as_vec3(field(v, "bitangent", "plane"), "plane.bitangent");
            nd.material_id = materials.get_id_or_insert(as_string(field(v, "material", "plane"), "plane.material"));
            scene.nodes.push_back(nd);
        } else {
            bad("unknown scene node type `" + type + "`");
        }
        (void)me;
    }

    // material/base.rs:25-48,66-78; material/pbr.rs:18-35,90-96;
    // material_input/mod.rs:15-39; material_input/texture.rs:17-43
    PtMaterialDesc parse_material(const std::string& text) {
        Value v = ptjson::parse(text);
        const std::string& type = as_string(field(v, "type", "material"), "material type");
        PtMaterialDesc m;
        memset(&m, 0, sizeof(m));
        // defaults of both variants
        m.color = {1.0f, 1.0f, 1.0f};
        m.emission = {1.0f, 1.0f, 1.0f};
        m.refraction = 1.0f;
        m.albedo = {1.0f, 1.0f, 1.0f};
        m.image_id = PT_NO_INDEX;
        if (type == "base") {
            m.type = PT_MATERIAL_BASE;
            if (const Value* c = v.find("color")) {
                const std::string& ct = as_string(field(*c, "type", "material color"), "material color type");
                if (ct == "color") {
                    m.color = as_vec3(field(*c, "color", "material color"), "material color");
                } else if (ct == "texture") {
                    m.color_is_texture = 1;
                    const std::string& path = as_string(field(*c, "path", "texture"), "texture.path");
                    const std::string& mode = as_string(field(*c, "uv_mode", "texture"), "texture.uv_mode");
                    if (mode == "clamp") m.uv_mode = PT_UV_CLAMP;
                    else if (mode == "repeat") m.uv_mode = PT_UV_REPEAT;
                    else bad("texture.uv_mode: unknown variant `" + mode + "`");
                    m.image_id = images.get_id_or_insert(path);
                } else {
                    bad("material color: unknown variant `" + ct + "`");
                }
            }
            if (const Value* e = v.find("emission")) m.emission = as_vec3(*e, "emission");
            if (const Value* e = v.find("refraction")) m.refraction = as_f32(*e, "refraction");
            if (const Value* e = v.find("reflective")) m.reflective = as_f32(*e, "reflective");
            if (const Value* e = v.find("emissive")) m.emissive = as_f32(*e, "emissive");
            if (const Value* e = v.find("refractive")) m.refractive = as_f32(*e, "refractive");
        } else if (type == "pbr") {
            m.type = PT_MATERIAL_PBR;
            if (const Value* e = v.find("albedo")) m.albedo = as_vec3(*e, "albedo");
            if (const Value* e = v.find("roughness")) m.roughness = as_f32(*e, "roughness");
            if (const Value* e = v.find("metallic")) m.metallic = as_f32(*e, "metallic");
        } else {
            bad("unknown material type `" + type + "`");
        }
        return m;
    }
};

} // namespace

void HostScene::finalize() {
    mesh_descs.resize(meshes.size());
    for (size_t i = 0; i < meshes.size(); i++) {
        const HostMesh& m = meshes[i];
        mesh_descs[i].triangle_count = m.triangle_count;
        mesh_descs[i].positions = m.positions.data();
        mesh_descs[i].normals = m.normals.empty() ? nullptr : m.normals.data();
        mesh_descs[i].normal_valid = m.normal_valid.empty() ? nullptr : m.normal_valid.data();
        mesh_descs[i].uvs = m.uvs.empty() ? nullptr : m.uvs.data();
    }
    image_descs.resize(images.size());
    for (size_t i = 0; i < images.size(); i++)
        image_descs[i] = {images[i].width, images[i].height, images[i].rgba8.data()};
    tree_blobs.resize(trees.size());
    for (size_t i = 0; i < trees.size(); i++) tree_blobs[i] = {trees[i].data(), (uint64_t)trees[i].size()};
    desc.nodes = nodes.data();
    desc.node_count = (uint32_t)nodes.size();
    desc.meshes = mesh_descs.data();
    desc.mesh_count = (uint32_t)mesh_descs.size();
    desc.materials = materials.data();
    desc.material_count = (uint32_t)materials.size();
    desc.images = image_descs.data();
    desc.image_count = (uint32_t)image_descs.size();
    desc.trees = tree_blobs.data();
    desc.tree_count = (uint32_t)tree_blobs.size();
}

std::unique_ptr<HostScene> load_scene(const FileStore& store, const std::string& scene_path) {
    auto scene = std::make_unique<HostScene>();
    Loader L{store, *scene, {}, {}, {}, {}};
    {
        std::vector<uint8_t> raw = store.fetch_file(scene_path);
        Value root = ptjson::parse(std::string(raw.begin(), raw.end()));
        L.add_node(root, 0);
    }
    // scene/mod.rs:71-99.  The reference walks its pending set in HashMap order;
    // ids are already fixed, so processing in id order gives the same scene, with
    // one difference: image ids (assigned while materials are initialised) come
    // out in material-id order here instead of an arbitrary order.
    for (const std::string& p : L.meshes.paths) {
        std::vector<uint8_t> raw = store.fetch_file(p);
        scene->meshes.push_back(load_obj(raw.data(), raw.size()));
    }
    for (const std::string& p : L.materials.paths) {
        std::vector<uint8_t> raw = store.fetch_file(p);
        scene->materials.push_back(L.parse_material(std::string(raw.begin(), raw.end())));
    }
    for (const std::string& p : L.images.paths) {
        std::vector<uint8_t> raw = store.fetch_file(p);
        scene->images.push_back(decode_png(raw.data(), raw.size()));
    }
    // ResourceType::KdTree is todo!() upstream (scene/mod.rs:94-96).  Here the file
    // is optional: missing or empty means "build the tree".
    for (const std::string& p : L.trees.paths) {
        if (store.exists(p)) scene->trees.push_back(store.fetch_file(p));
        else scene->trees.emplace_back();
    }
    scene->mesh_paths = L.meshes.paths;
    scene->material_paths = L.materials.paths;
    scene->image_paths = L.images.paths;
    scene->tree_paths = L.trees.paths;
    scene->finalize();
    return scene;
}

void parse_render_task(const std::string& json_text, PtRenderTask* out, std::string* scene_path, std::string* scene_md5) {
    Value v = ptjson::parse(json_text);
    memset(out, 0, sizeof(*out));
    if (scene_path) *scene_path = as_string(field(v, "scene", "render task"), "render task scene");
    if (scene_md5) {
        const Value* md5 = v.kind == Value::Object ? v.find("scene_md5") : nullptr;
        *scene_md5 = md5 ? as_string(*md5, "render task scene_md5") : std::string();
    }
    const Value& config = field(v, "config", "render task");
    out->trace_depth = as_usize(field(config, "trace_depth", "config"), "config.trace_depth");
    out->iterations = as_usize(field(config, "iterations", "config"), "config.iterations");
    const Value& cam = field(v, "camera", "render task");
    const Value& res = field(cam, "resolution", "camera");
    if (res.kind != Value::Array || res.arr.size() != 2) bad("camera.resolution: expecting [<x>, <y>] array");
    PtCamera& c = out->camera;
    c.resolution_x = as_usize(res.arr[0], "camera.resolution");
    c.resolution_y = as_usize(res.arr[1], "camera.resolution");
    c.rotation = as_mat3(field(cam, "rotation", "camera"), "camera.rotation");
    c.position = as_vec3(field(cam, "position", "camera"), "camera.position");
    c.fov = as_f32(field(cam, "field_of_view", "camera"), "camera.field_of_view");
    c.near_plane = as_f32(field(cam, "near_plane", "camera"), "camera.near_plane");
    c.focal_length = as_f32(field(cam, "focal_length", "camera"), "camera.focal_length");
    const std::string& shape = as_string(field(cam, "bokeh_shape", "camera"), "camera.bokeh_shape");
    if (shape == "point") c.bokeh_shape = PT_BOKEH_POINT;
    else if (shape == "circle") c.bokeh_shape = PT_BOKEH_CIRCLE;
    else if (shape == "square") c.bokeh_shape = PT_BOKEH_SQUARE;
    else bad("camera.bokeh_shape: unknown variant `" + shape + "`");
    c.bokeh_size = as_f32(field(cam, "bokeh_size", "camera"), "camera.bokeh_size");
}

} // namespace pt
