// C++ host mirror of the reference worker's rendering interface, over the C ABI
// of pathtracing_amd.h (header only; link with -lpathtracing_amd).
//
//   pt::Scene        Arc<Scene>       worker/src/scene/mod.rs:27-112  (Scene::load)
//   pt::RenderTask   RenderTask       worker/src/api/render_task.rs:23-49
//   pt::HipRenderer  impl Renderer    worker/src/renderer/mod.rs:9-13 (init / render)
//   pt::RenderedImage                 worker/src/lib.rs:51-79         (to_bytes / from_bytes)
//   pt::Frame        Frame            client/src/frame.rs:11-103      (sum of renders, mean, RGBA8 preview)
//   pt::Worker       Worker           worker/src/lib.rs:20-48         (scene cache + render)
//
// Where the reference panics (missing file, malformed JSON, ...) these throw pt::Failure.
#ifndef PATHTRACING_AMD_HPP
#define PATHTRACING_AMD_HPP

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pathtracing_amd.h"

namespace pt {

class Failure : public std::runtime_error {
public:
    Failure(PtStatus status, const std::string& what) : std::runtime_error(what), status_(status) {}
    PtStatus status() const { return status_; }

private:
    PtStatus status_;
};

inline void check(PtStatus status) {
    if (status == PT_OK) return;
    std::string msg(pt_last_error(nullptr, 0) + 1, '\0');
    pt_last_error(&msg[0], msg.size());
    msg.resize(msg.size() - 1);
    throw Failure(status, msg);
}

// Scene::load: the scene JSON and every resource it references, from a local directory.
class Scene {
public:
    static std::shared_ptr<Scene> load(const std::string& root_dir, const std::string& scene_path) {
        PtHostScene* h = nullptr;
        check(pt_host_scene_load(root_dir.c_str(), scene_path.c_str(), &h));
        return std::shared_ptr<Scene>(new Scene(h));
    }
    ~Scene() { pt_host_scene_destroy(handle_); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    const PtSceneDesc* desc() const { return pt_host_scene_desc(handle_); }
    // Writes the .tree file of every kd_tree node whose file is missing or does not fit its mesh and takes
    // it into this scene (kd_tree.rs:12-51 names the file; nothing upstream writes it). Returns how many.
    uint32_t cache_trees(const std::string& root_dir) {
        uint32_t written = 0;
        check(pt_host_scene_cache_trees(handle_, root_dir.c_str(), &written));
        return written;
    }

private:
    explicit Scene(PtHostScene* h) : handle_(h) {}
    PtHostScene* handle_;
};

struct RenderTask {
    std::string scene; // path of the scene JSON, as in the task
    PtRenderTask task{};

    static RenderTask from_json(const std::string& text) {
        RenderTask t;
        char path[4096];
        check(pt_render_task_parse(text.c_str(), &t.task, path, sizeof(path)));
        t.scene = path;
        return t;
    }
};

// Rgb32FImage + the bytes the worker sends back.
struct RenderedImage {
    uint32_t width = 0, height = 0;
    std::vector<float> rgb; // row-major, top row first, 3 floats per pixel

    std::vector<uint8_t> to_bytes() const {
        std::vector<uint8_t> out(8 + rgb.size() * sizeof(float));
        check(pt_rendered_image_to_bytes(rgb.data(), width, height, out.data(), out.size()));
        return out;
    }

    static RenderedImage from_bytes(const std::vector<uint8_t>& bytes) {
        RenderedImage img;
        check(pt_rendered_image_from_bytes(bytes.data(), bytes.size(), &img.width, &img.height, nullptr, 0));
        img.rgb.resize((size_t)img.width * img.height * 3);
        check(pt_rendered_image_from_bytes(bytes.data(), bytes.size(), &img.width, &img.height, img.rgb.data(), img.rgb.size()));
        return img;
    }
};

// Frame (client/src/frame.rs:11-103): the running sum of the renders of one frame on one GPU,
// its mean image and the RGBA8 preview the client shows.
class Frame {
public:
    Frame(uint32_t width, uint32_t height, int device = 0) : width_(width), height_(height) {
        check(pt_frame_create(width, height, device, &handle_));
    }
    Frame(const Frame&) = delete;
    Frame& operator=(const Frame&) = delete;
    ~Frame() { pt_frame_destroy(handle_); }

    void add_render(const RenderedImage& render) {
        if (render.width != width_ || render.height != height_) throw Failure(PT_ERR_INVALID_ARGUMENT, "render and frame differ in size");
        check(pt_frame_add_render(handle_, render.rgb.data()));
    }
    uint64_t count() const { return pt_frame_render_count(handle_); }
    RenderedImage mean_image() const {
        RenderedImage img;
        img.width = width_;
        img.height = height_;
        img.rgb.resize((size_t)width_ * height_ * 3);
        check(pt_frame_image(handle_, img.rgb.data(), nullptr));
        return img;
    }
    std::vector<uint8_t> image_rgba8() const { // gamma_correction, frame.rs:86-103
        std::vector<uint8_t> out((size_t)width_ * height_ * 4);
        check(pt_frame_image(handle_, nullptr, out.data()));
        return out;
    }

private:
    PtFrame* handle_ = nullptr;
    uint32_t width_, height_;
};

// trait Renderer { fn init(scene: Arc<Scene>) -> Self; async fn render(&mut self, task) -> Rgb32FImage }
class HipRenderer {
public:
    static HipRenderer init(std::shared_ptr<const Scene> scene, int device = 0) {
        PtScene* h = nullptr;
        check(pt_scene_create(scene->desc(), device, &h));
        return HipRenderer(h, std::move(scene));
    }
    HipRenderer(HipRenderer&& o) noexcept : handle_(o.handle_), scene_(std::move(o.scene_)) { o.handle_ = nullptr; }
    HipRenderer& operator=(HipRenderer&& o) noexcept {
        if (this != &o) {
            pt_scene_destroy(handle_);
            handle_ = o.handle_;
            scene_ = std::move(o.scene_);
            o.handle_ = nullptr;
        }
        return *this;
    }
    HipRenderer(const HipRenderer&) = delete;
    HipRenderer& operator=(const HipRenderer&) = delete;
    ~HipRenderer() { pt_scene_destroy(handle_); }

    RenderedImage render(const RenderTask& task, const PtRenderOpts* opts = nullptr, PtRenderStats* stats = nullptr) {
        RenderedImage img;
        img.width = (uint32_t)task.task.camera.resolution_x;
        img.height = (uint32_t)task.task.camera.resolution_y;
        img.rgb.resize((size_t)img.width * img.height * 3);
        check(pt_render(handle_, &task.task, opts, img.rgb.data(), stats));
        return img;
    }
    // What pt_scene_create did: .tree files offered / adopted, structures built, times.
    PtSceneInfo info() const {
        PtSceneInfo out{};
        check(pt_scene_info(handle_, &out));
        return out;
    }
    PtScene* handle() const { return handle_; }

private:
    HipRenderer(PtScene* h, std::shared_ptr<const Scene> s) : handle_(h), scene_(std::move(s)) {}
    PtScene* handle_;
    std::shared_ptr<const Scene> scene_;
};

// Worker::render (lib.rs:31-48): loads the task's scene unless it is the one already held, renders.
// The reference keys its cache on scene_md5; a local directory has no md5, so the scene path is the key,
// and the renderer (BVH + device copies) is kept with the scene instead of being rebuilt per task.
class Worker {
public:
    explicit Worker(std::string root_dir, int device = 0) : root_(std::move(root_dir)), device_(device) {}

    RenderedImage render(const RenderTask& task, const PtRenderOpts* opts = nullptr, PtRenderStats* stats = nullptr) {
        if (!renderer_ || task.scene != scene_path_) {
            renderer_.reset();
            auto scene = Scene::load(root_, task.scene);
            renderer_.reset(new HipRenderer(HipRenderer::init(scene, device_)));
            scene_path_ = task.scene;
        }
        return renderer_->render(task, opts, stats);
    }

private:
    std::string root_, scene_path_;
    int device_;
    std::unique_ptr<HipRenderer> renderer_;
};

// One task on several GPUs of this process: one renderer per device from the same scene, the task split by
// sample range (PT_SHARD_SAMPLES) or bands of 8 pixel rows (PT_SHARD_ROWS), the partial sums reduced onto the
// first device.  What worker_pool.rs:221-236 + frame.rs:60-84 do across workers, inside one worker.
class MultiRenderer {
public:
    static MultiRenderer init(std::shared_ptr<const Scene> scene, const std::vector<int>& devices) {
        MultiRenderer m;
        for (int d : devices) m.renderers_.push_back(HipRenderer::init(scene, d));
        return m;
    }
    RenderedImage render(const RenderTask& task, PtShardMode mode = PT_SHARD_SAMPLES, const PtRenderOpts* opts = nullptr,
                         PtRenderStats* stats = nullptr) {
        std::vector<PtScene*> handles;
        for (const HipRenderer& r : renderers_) handles.push_back(r.handle());
        RenderedImage img;
        img.width = (uint32_t)task.task.camera.resolution_x;
        img.height = (uint32_t)task.task.camera.resolution_y;
        img.rgb.resize((size_t)img.width * img.height * 3);
        check(pt_render_multi(handles.data(), (int)handles.size(), &task.task, opts, (uint32_t)mode, img.rgb.data(), stats));
        return img;
    }

private:
    std::vector<HipRenderer> renderers_;
};

// The worker's front end without its sockets (worker/src/main.rs:36-113): the discovery reply and one round of
// connection_loop over a store of directories named by scene_md5 (file_store.rs:17-23).
inline std::vector<uint8_t> discovery_response(uint16_t websocket_port) { // lib.rs:86-91, main.rs:104-110
    uint8_t buf[8];
    size_t n = 0;
    check(pt_discovery_response_encode(websocket_port, buf, sizeof(buf), &n));
    return std::vector<uint8_t>(buf, buf + n);
}
inline bool is_discovery_request(const uint8_t* datagram, size_t size) { // lib.rs:83-84: `Request {}` is the empty message
    return pt_discovery_request_decode(datagram, size) == PT_OK;
}
inline std::string scene_md5(const std::string& root_dir, const std::string& scene_path) { // client/src/scene.rs:19-85
    char hex[33];
    check(pt_scene_md5(root_dir.c_str(), scene_path.c_str(), hex));
    return std::string(hex);
}
class WorkerFrontEnd {
public:
    explicit WorkerFrontEnd(const std::string& store_root, int device = 0) { check(pt_worker_create(store_root.c_str(), device, &handle_)); }
    WorkerFrontEnd(const WorkerFrontEnd&) = delete;
    WorkerFrontEnd& operator=(const WorkerFrontEnd&) = delete;
    ~WorkerFrontEnd() { pt_worker_destroy(handle_); }
    // main.rs:68-93: a text frame holding a RenderTask in, the RenderedImage bytes of the reply out
    std::vector<uint8_t> handle_message(const std::string& task_json, const PtRenderOpts* opts = nullptr) {
        uint8_t* bytes = nullptr;
        uint64_t size = 0;
        check(pt_worker_handle_message(handle_, task_json.c_str(), opts, &bytes, &size));
        std::vector<uint8_t> out(bytes, bytes + size);
        pt_buffer_free(bytes);
        return out;
    }
    uint32_t cached_scenes() const { return pt_worker_cached_scene_count(handle_); }
    PtWorker* handle() const { return handle_; }

private:
    PtWorker* handle_ = nullptr;
};

// start_ws (main.rs:36-48): the WebSocket listener and the UDP discovery responder over a worker, served by the
// library's own threads until the object goes away.  Port 0 = any free port.  The worker must outlive it.
class Server {
public:
    Server(WorkerFrontEnd& worker, uint16_t websocket_port = 30000, uint16_t udp_port = 40000, const std::string& bind_addr = "0.0.0.0") {
        check(pt_worker_serve_start(worker.handle(), bind_addr.c_str(), websocket_port, udp_port, &handle_));
        check(pt_server_ports(handle_, &websocket_port_, &udp_port_));
    }
    Server(const Server&) = delete;
    Server& operator=(const Server&) = delete;
    ~Server() { pt_server_stop(handle_); }
    uint16_t websocket_port() const { return websocket_port_; }
    uint16_t udp_port() const { return udp_port_; }
    PtServerStats stats(std::string* last_error = nullptr) const {
        PtServerStats st{};
        char message[512] = {0};
        check(pt_server_stats(handle_, &st, message, sizeof(message)));
        if (last_error) *last_error = message;
        return st;
    }

private:
    PtServer* handle_ = nullptr;
    uint16_t websocket_port_ = 0, udp_port_ = 0;
};

} // namespace pt
#endif
