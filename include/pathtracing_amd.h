/*
 * pathtracing_amd.h -- C ABI of the MI355X (gfx950) path-tracing core.
 *
 * This is the drop-in boundary for the per-pixel render loop of the
 * mertwole/PathTracing `worker` crate.  Every entry point is `extern "C"`,
 * takes plain pointers and sizes, never throws and never aborts across the
 * boundary: failures come back as a non-zero PtStatus and a message readable
 * through pt_last_error().
 *
 * Reference interface replaced (paths relative to the reference checkout):
 *   trait Renderer { fn init(scene: Arc<Scene>) -> Self;
 *                    async fn render(&mut self, Arc<RenderTask>) -> Rgb32FImage }
 *                                           worker/src/renderer/mod.rs:9-13
 *   its only implementor CPURenderer       worker/src/renderer/cpu_renderer/mod.rs:178-195
 *   its only caller Worker::render          worker/src/lib.rs:31-48
 *
 * The structs below are the #[repr(C)] mirrors of the reference's plain f32
 * types (math/src/lib.rs: Vec2 :118-123, Vec3 :214-220, Mat3 :369-375,
 * Mat4 :544-551; worker/src/ray.rs:3-9; worker/src/camera.rs:35-48).
 */
#ifndef PATHTRACING_AMD_H
#define PATHTRACING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 3u

/* ------------------------------------------------------------------ status */

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = 1, /* null pointer, bad enum, inconsistent counts        */
    PT_ERR_SCENE = 2,            /* scene description rejected (e.g. plane axes not
                                    orthogonal: the reference asserts, plane.rs:49-51) */
    PT_ERR_IO = 3,               /* file missing / unreadable                           */
    PT_ERR_PARSE = 4,            /* malformed JSON / OBJ / PNG / .tree                  */
    PT_ERR_DEVICE = 5,           /* HIP runtime error, no device, out of memory         */
    PT_ERR_UNSUPPORTED = 6       /* valid input the library does not handle             */
} PtStatus;

/* Copies the calling thread's last error message (NUL terminated, truncated to
 * cap) into buf and returns the untruncated length.  Replaces the reference's
 * panics (unwrap/expect all over worker/src). */
size_t pt_last_error(char* buf, size_t cap);

/* Library / ABI version: returns PT_ABI_VERSION of the built library. */
uint32_t pt_abi_version(void);

/* ------------------------------------------------------- math-type mirrors */

typedef struct PtVec2 { float x, y; } PtVec2;                 /* math Vec2 */
typedef struct PtVec3 { float x, y, z; } PtVec3;              /* math Vec3 */
typedef struct PtMat3 { PtVec3 row0, row1, row2; } PtMat3;    /* math Mat3, row-major */
typedef struct PtMat4 { float m[4][4]; } PtMat4;              /* math Mat4, m[row][col] */

/* worker/src/ray.rs:3-9 */
typedef struct PtRay {
    PtVec3 source;
    PtVec3 direction;
    float min;
    float max;
} PtRay;

/* --------------------------------------------------------------- the scene */

/* Scene-graph node kinds; JSON "type" tags of the files in worker/src/scene/hierarchy/ */
typedef enum PtNodeType {
    PT_NODE_COLLECTION = 0, /* node_collection.rs:15-18 */
    PT_NODE_TRANSFORM = 1,  /* transform.rs:17-25       */
    PT_NODE_MESH = 2,       /* hierarchy/mesh.rs:18-22  */
    PT_NODE_SPHERE = 3,     /* sphere.rs:20-29          */
    PT_NODE_PLANE = 4,      /* plane.rs:20-28           */
    PT_NODE_KD_TREE = 5     /* kd_tree.rs:15-19 (trace_ray is todo!() upstream; here it
                               is "the child, accelerated": identical result to the child) */
} PtNodeType;

#define PT_NO_INDEX 0xFFFFFFFFu

/* One scene-graph node.  Nodes are stored in depth-first pre-order: a node is
 * followed by its whole first child subtree, then the next child, ...  Only the
 * fields of the node's own type are read. */
typedef struct PtNodeDesc {
    uint32_t type;        /* PtNodeType */
    uint32_t child_count; /* COLLECTION: number of children; TRANSFORM, KD_TREE: 1; leaves: 0 */
    /* TRANSFORM */
    PtMat4 matrix;
    /* MESH */
    uint32_t mesh_id;     /* index into PtSceneDesc.meshes (Scene.meshes, scene/mod.rs:44) */
    /* MESH, SPHERE, PLANE */
    uint32_t material_id; /* index into PtSceneDesc.materials */
    /* SPHERE */
    PtVec3 center;
    float radius;
    /* PLANE: as written in the JSON; normalised and checked for orthogonality by
     * pt_scene_create exactly as plane.rs:45-51 does */
    PtVec3 point, normal, tangent, bitangent;
    /* KD_TREE: index into PtSceneDesc.trees or PT_NO_INDEX (tree is then built) */
    uint32_t tree_id;
} PtNodeDesc;

/* A triangle mesh resource = resource::Mesh (mesh/mod.rs:19-24) before
 * TriangleUninit::init: triangles in OBJ face order (the order is the tie-break
 * order of the closest-hit search, mesh/mod.rs:57-62). */
typedef struct PtMeshDesc {
    uint64_t triangle_count;
    const float* positions;       /* 9 floats per triangle: v0.xyz v1.xyz v2.xyz            */
    const float* normals;         /* 9 floats per triangle, or NULL: none given              */
    const uint8_t* normal_valid;  /* 1 byte per triangle (0: use the face normal), or NULL:
                                     every triangle of `normals` is valid                    */
    const float* uvs;             /* 6 floats per triangle, or NULL: (0,0) everywhere        */
} PtMeshDesc;

typedef enum PtMaterialType { PT_MATERIAL_BASE = 0, PT_MATERIAL_PBR = 1 } PtMaterialType;
typedef enum PtUvMode { PT_UV_CLAMP = 0, PT_UV_REPEAT = 1 } PtUvMode;

/* material/base.rs:25-48 and material/pbr.rs:18-35, already defaulted. */
typedef struct PtMaterialDesc {
    uint32_t type; /* PtMaterialType */
    /* base */
    uint32_t color_is_texture; /* 0: `color`, 1: texture (image_id, uv_mode) */
    PtVec3 color;
    uint32_t image_id;
    uint32_t uv_mode; /* PtUvMode */
    PtVec3 emission;
    float refraction, reflective, emissive, refractive;
    /* pbr */
    PtVec3 albedo;
    float roughness, metallic;
} PtMaterialDesc;

/* resource/image.rs: RGBA8, row-major, top row first. */
typedef struct PtImageDesc {
    uint32_t width, height;
    const uint8_t* rgba8;
} PtImageDesc;

/* Serialized acceleration structure (".tree" file contents) for one mesh. */
typedef struct PtTreeBlob {
    const uint8_t* data;
    uint64_t size; /* 0: empty file, build instead (scene_data/cube.tree is 0 bytes) */
} PtTreeBlob;

typedef struct PtSceneDesc {
    const PtNodeDesc* nodes;
    uint32_t node_count; /* nodes[0] is the root (Scene.hierarchy) */
    const PtMeshDesc* meshes;
    uint32_t mesh_count;
    const PtMaterialDesc* materials;
    uint32_t material_count;
    const PtImageDesc* images;
    uint32_t image_count;
    const PtTreeBlob* trees;
    uint32_t tree_count;
} PtSceneDesc;

typedef struct PtScene PtScene; /* opaque: device-resident scene */

/* Renderer::init (renderer/mod.rs:10): validates the description, builds (or
 * adopts) the acceleration structures, uploads everything to `device`.  All
 * input arrays are copied; the caller keeps ownership. */
PtStatus pt_scene_create(const PtSceneDesc* desc, int device, PtScene** out_scene);
void pt_scene_destroy(PtScene* scene);

/* What pt_scene_create did with the scene's meshes.  A kd_tree node directly above a mesh
 * node offers its .tree blob to that mesh (kd_tree.rs:12-51 names the file; loading it is
 * todo!() upstream, scene/mod.rs:94-96): a blob that is well formed for exactly this
 * geometry is adopted and the build is skipped, any other blob (empty, stale, corrupt) is
 * ignored and the structure is built. */
typedef struct PtSceneInfo {
    uint32_t mesh_count;
    uint32_t trees_offered; /* meshes a non-empty .tree blob was offered for                 */
    uint32_t trees_adopted; /* ... whose blob was adopted: the GPU searches the file's tree   */
    uint32_t trees_built;   /* meshes whose structure was built by pt_scene_create            */
    uint32_t instance_run_children; /* children of the root collection taken as one instance run */
    uint64_t triangle_count; /* over all meshes                                               */
    uint64_t bvh_bytes;      /* device bytes of nodes + leaf triangles of all meshes          */
    double build_ms;         /* host time spent building (not adopting) structures            */
    double setup_ms;         /* the whole of pt_scene_create                                  */
} PtSceneInfo;
PtStatus pt_scene_info(const PtScene* scene, PtSceneInfo* out_info);

/* The queues of the wavefront pipeline belong to the device, not to a scene: all scenes of
 * a device render through the same workspace (two when a render keeps two batches in
 * flight), sized by the largest demand so far.  Calls on any scene and any stream of one
 * device are ordered on the GPU by events, so a PtScene may be used from several streams
 * and threads; renders of one device do not overlap each other.  This frees that memory
 * (it is allocated again on demand); it waits for the device's queued renders first. */
PtStatus pt_device_release_workspace(int device);

/* ---------------------------------------------------------- the render task */

typedef enum PtBokehShape { PT_BOKEH_POINT = 0, PT_BOKEH_CIRCLE = 1, PT_BOKEH_SQUARE = 2 } PtBokehShape;

/* worker/src/camera.rs:35-48 */
typedef struct PtCamera {
    uint64_t resolution_x, resolution_y;
    PtMat3 rotation;
    PtVec3 position;
    float fov; /* JSON field_of_view */
    float near_plane;
    float focal_length;
    uint32_t bokeh_shape; /* PtBokehShape */
    float bokeh_size;
} PtCamera;

/* worker/src/api/render_task.rs:23-29,45-49 minus the two strings */
typedef struct PtRenderTask {
    PtCamera camera;
    uint64_t trace_depth;
    uint64_t iterations;
} PtRenderTask;

#define PT_FLAG_BRUTE_FORCE 1u /* scan every triangle like mesh/mod.rs:57-62 (no BVH) */
#define PT_FLAG_PROFILE 2u     /* time every launch of the BVH search kernel with events on the
                                  render stream (fills PtRenderStats.search_kernel_*); slower  */
#define PT_FLAG_WAVEFRONT 4u   /* pt_trace_closest: take the rays through the kernels a render
                                  uses (queue regions, listed searches, search kernel, walk)
                                  instead of one thread per ray; rays must have max = f32::MAX */
#define PT_FLAG_WAVEFRONT_SPLIT 8u /* the same, with the separate walk kernel even where a
                                  render would use the fused walk + bounce kernel              */
#define PT_FLAG_SINGLE_STREAM 16u /* render: never keep two batches in flight                  */
#define PT_FLAG_COUNT_VISITS 32u /* render: the BVH search kernel counts the node and triangle
                                  records it fetches (PtRenderStats.node_visits,
                                  .triangle_tests); a little slower                            */

/* Knobs the reference does not have.  A NULL pointer means all zero. */
typedef struct PtRenderOpts {
    uint64_t seed;          /* key of the counter-based random stream                      */
    uint64_t sample_offset; /* first sample index rendered (sample-range sharding)          */
    uint32_t flags;         /* PT_FLAG_*                                                    */
    uint32_t tile_x0, tile_y0, tile_w, tile_h; /* camera-space pixel rectangle to render
                               (y up); w==0 or h==0: whole frame.  Pixels outside stay 0.   */
    uint32_t paths_per_batch; /* path slots traced at a time (whole samples of a band of 8-pixel
                               rows).  0: library default -- up to 64 Mi (16 Mi for scenes with an
                               instance run), in batches of equal size, at least two of them when
                               the render has 8 Mi paths or more (two are kept in flight).  The
                               image does not depend on it.                                    */
} PtRenderOpts;

typedef struct PtRenderStats {
    uint64_t rays;        /* path segments traced = iterations of work_group.rs:29-47 */
    uint64_t paths;
    double render_ms;     /* device time of the render itself                          */
    uint64_t bvh_bytes;   /* device bytes of nodes + triangles of all meshes           */
    uint64_t searches;    /* BVH searches run (rays that reached a mesh's bounds)        */
    double search_kernel_ms;         /* PT_FLAG_PROFILE: summed duration of the BVH search
                                        kernel launches                                        */
    uint64_t search_kernel_launches; /* PT_FLAG_PROFILE: how many launches that sum covers    */
    uint64_t node_visits;    /* PT_FLAG_COUNT_VISITS: 64-byte node records the searches fetched */
    uint64_t triangle_tests; /* PT_FLAG_COUNT_VISITS: 48-byte triangle records they tested      */
    /* PT_FLAG_COUNT_VISITS, per wave of 64 lanes: node steps taken (each serves the lanes that are at
     * an inner node), leaf steps taken, and the triangle rounds inside them (a leaf step lasts as
     * many rounds as its fullest leaf has triangles): node_visits / (64 * wave_node_steps) is the
     * share of lanes at work in a node step */
    uint64_t wave_node_steps, wave_leaf_steps, wave_leaf_rounds;
    double reduce_ms;     /* pt_render_multi: host time of the sum onto the first device (0 elsewhere) */
} PtRenderStats;

/* Renderer::render (renderer/mod.rs:12) -> Rgb32FImage: fills out_rgb
 * (resolution_x * resolution_y * 3 floats, row-major, top row first, i.e.
 * already flipped like get_image, cpu_renderer/mod.rs:143-175) with
 * sum_of_samples * (1 / iterations).  out_rgb is host memory. */
PtStatus pt_render(PtScene* scene, const PtRenderTask* task, const PtRenderOpts* opts,
                   float* out_rgb, PtRenderStats* stats);

/* Same render, for callers that keep buffers on the GPU (sample-range sharding
 * followed by a collective sum).  d_accum (device memory, resolution_x *
 * resolution_y * 3 floats, camera-space layout: pixel (x,y) at (x + y*W)*3, y
 * up) receives  d_accum += sum over this call's samples; nothing is normalised
 * or flipped.  `stream` is a hipStream_t (NULL: the default stream); the call
 * returns after enqueueing unless stats != NULL (then it synchronises).  A render
 * of more than one batch traces every other batch on a stream of the library's
 * own; all of it is ordered before whatever is queued on `stream` afterwards. */
PtStatus pt_render_accumulate_device(PtScene* scene, const PtRenderTask* task,
                                     const PtRenderOpts* opts, float* d_accum, void* stream,
                                     PtRenderStats* stats);

/* get_image (cpu_renderer/mod.rs:143-175) on the device: d_out_rgb[(x + (H-1-y)*W)*3+c]
 * = d_accum[(x + y*W)*3+c] * (1.0f / total_iterations). */
PtStatus pt_resolve_device(const float* d_accum, uint64_t width, uint64_t height,
                           uint64_t total_iterations, float* d_out_rgb, void* stream);

/* ------------------------------------------------- several GPUs, one process */

/* How a render is cut over ranks or devices.  The reference combines workers on the host:
 * the client sends the same task to every worker and averages the frames
 * (client/src/worker_pool.rs:221-236, client/src/frame.rs:60-84). */
typedef enum PtShardMode {
    PT_SHARD_SAMPLES = 0, /* disjoint sample ranges of every pixel (sums differ from one GPU's in f32 order only) */
    PT_SHARD_ROWS = 1     /* bands of 8-pixel rows, every sample (bit-identical to one GPU's image)               */
} PtShardMode;

/* The share of `rank` of `world`: out_task / out_opts are `task` / `opts` (NULL: zeros) with
 * the iterations and sample_offset of the rank's sample range, or the tile of its row band;
 * *out_active = 0 when the rank has nothing to render. */
PtStatus pt_shard(const PtRenderTask* task, const PtRenderOpts* opts, uint32_t mode, uint32_t world, uint32_t rank,
                  PtRenderTask* out_task, PtRenderOpts* out_opts, uint32_t* out_active);

/* Renderer::render on n_devices GPUs of one node from one process: scenes[i] is the scene
 * resident on the i-th device (pt_scene_create of the same description on each); every
 * device renders its share (pt_shard, on a stream and host thread of its own), the
 * un-normalised sums are added onto the first device -- one grouped RCCL reduce(sum) over
 * xGMI when the devices are distinct and RCCL can be opened, peer copies and adds
 * otherwise (and with PATHTRACING_AMD_REDUCE=p2p) -- which resolves the image into out_rgb
 * (layout as pt_render; host memory, or memory of scenes[0]'s device, which saves the copy
 * back).  n_devices = 1 is pt_render.  task->iterations = 0, and a tile in `opts` together
 * with PT_SHARD_ROWS, are refused.  Streams, the per-device accumulation buffers and the
 * communicators are kept per list of devices for later calls (pt_device_release_workspace
 * frees the buffers); calls naming the same list of devices take turns.  stats (may be
 * NULL): counters summed over the devices, render_ms = the call on the host's clock without
 * that one-time set-up, reduce_ms = the part of it spent on the sum. */
PtStatus pt_render_multi(PtScene** scenes, int n_devices, const PtRenderTask* task, const PtRenderOpts* opts,
                         uint32_t mode, float* out_rgb, PtRenderStats* stats);

/* ----------------------------------------------------- closest-hit contract */

typedef struct PtHit {
    uint32_t hit;        /* RayTraceResult.hit */
    uint32_t hit_inside; /* RayTraceResult.hit_inside */
    uint32_t node_index; /* index of the leaf node (mesh/sphere/plane) in PtSceneDesc.nodes */
    uint32_t tri_index;  /* position in the mesh's triangle list, PT_NO_INDEX for sphere/plane */
    uint32_t material_id;
    float t;
    PtVec3 point;
    PtVec3 normal;
    PtVec2 uv;
} PtHit;

/* scene.hierarchy.trace_ray(ray) for n rays (host arrays).  flags: PT_FLAG_BRUTE_FORCE,
 * PT_FLAG_WAVEFRONT, PT_FLAG_WAVEFRONT_SPLIT. */
PtStatus pt_trace_closest(PtScene* scene, const PtRay* rays, size_t n, uint32_t flags,
                          PtHit* out_hits);

/* pt_trace_closest for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart
 * in the reference.  d_rays and d_hits are device memory of the scene's device; the
 * kernel's scratch (the per-lane columns that take the overflow of the BVH traversal
 * stack and the saved frames of nested scene-graph nodes) is a static device array of
 * 19.4 MB that every GPU the library runs on holds until the process ends, searched on
 * a fixed grid of 64 blocks -- a quarter of the chip: for small, latency-bound queries.
 * The call allocates and frees nothing, enqueues one kernel on `stream` (NULL: the
 * default stream) and returns; n above PT_STATIC_TRACE_MAX_RAYS is
 * PT_ERR_INVALID_ARGUMENT.  flags: PT_FLAG_BRUTE_FORCE or 0.  All calls on one GPU share
 * the array: calls that overlap there (different streams, no ordering between them)
 * race; calls on one stream are ordered.  Nothing in the array has to be cleared between
 * calls (every entry is written before it is read). */
#define PT_STATIC_TRACE_MAX_RAYS 16777216u
PtStatus pt_trace_closest_device_static(PtScene* scene, const PtRay* d_rays, size_t n, uint32_t flags,
                                        PtHit* d_hits, void* stream);

/* Camera::get_ray (camera.rs:51-83) for `n` (pixel, sample) pairs, with the
 * library's counter-based random stream; host arrays.  pixel_xy holds n pairs
 * (x, y) in camera space. */
PtStatus pt_generate_rays(const PtCamera* camera, uint64_t seed, const uint32_t* pixel_xy,
                          const uint64_t* sample_index, size_t n, int device, PtRay* out_rays);

/* Material::get_color (renderer/cpu_renderer/mod.rs:50-57) for n (direction,
 * hit) pairs with the random stream of (seed, stream_id[i]) starting at draw
 * counter 0.  out_kind: 0 = Color, 1 = NextRayColorMultiplierAndDirection. */
PtStatus pt_shade(PtScene* scene, const PtVec3* directions, const PtHit* hits,
                  const uint64_t* stream_id, uint64_t seed, size_t n, uint32_t* out_kind,
                  PtVec3* out_color_or_multiplier, PtVec3* out_direction);

/* ------------------------------------------------- host-side scene loading */

typedef struct PtHostScene PtHostScene; /* opaque: parsed scene files in host memory */

/* Scene::load (scene/mod.rs:58-112) reading from a local directory instead of
 * GridFS (file_store.rs:27-35): `root_dir` + "/" + path as written in the JSON.
 * Resource ids are assigned in depth-first first-touch order like
 * ReferenceMapping (resource/mod.rs:28-34). */
PtStatus pt_host_scene_load(const char* root_dir, const char* scene_path, PtHostScene** out);
const PtSceneDesc* pt_host_scene_desc(const PtHostScene* scene);
void pt_host_scene_destroy(PtHostScene* scene);

/* The write side of the .tree cache: for every kd_tree node directly above a mesh node
 * whose .tree file is missing, empty or not valid for the mesh, builds the mesh's
 * acceleration structure, writes it to `root_dir` + "/" + the node's path (through a
 * temporary file and a rename) and puts it into the loaded scene, so that this and every
 * later pt_scene_create of the scene adopts it instead of building.  out_written (may be
 * NULL) receives the number of files written.  The reference has no counterpart: its
 * loader for this resource type is todo!() (scene/mod.rs:94-96). */
PtStatus pt_host_scene_cache_trees(PtHostScene* scene, const char* root_dir, uint32_t* out_written);

/* Parses a render task (RenderTaskUninit or RenderTask JSON text,
 * api/render_task.rs:5-29).  scene_path_buf receives the "scene" string. */
PtStatus pt_render_task_parse(const char* json_text, PtRenderTask* out_task,
                              char* scene_path_buf, size_t scene_path_cap);

/* ---------------------------------------------------------- .tree (de)ser. */

/* Builds the acceleration structure of one mesh and serialises it.  The buffer
 * is owned by the library: release with pt_buffer_free. */
PtStatus pt_tree_build(const PtMeshDesc* mesh, uint8_t** out_data, uint64_t* out_size);
void pt_buffer_free(uint8_t* data);

/* PT_OK when `blob` is a well-formed tree for exactly this mesh: right version and
 * geometry hash, every triangle in exactly one leaf, depth within the traversal
 * stack, every box enclosing the padded bounds of what lies below it.  A blob that
 * fails is never used: pt_scene_create rebuilds instead (PT_ERR_PARSE here). */
PtStatus pt_tree_validate(const PtMeshDesc* mesh, const uint8_t* data, uint64_t size);

/* ---------------------------------------------------------- result codecs */

/* RenderedImage::to_bytes (worker/src/lib.rs:56-62): u32 LE width, u32 LE height,
 * then width*height*3 f32 LE.  out must hold 8 + width*height*12 bytes. */
PtStatus pt_rendered_image_to_bytes(const float* rgb, uint32_t width, uint32_t height,
                                    uint8_t* out, size_t out_cap);

/* RenderedImage::from_bytes (worker/src/lib.rs:64-78), what the client does with a
 * worker's reply: width and height from the first 8 bytes, then the floats.
 * out_rgb may be NULL to read the size only; otherwise it receives
 * width*height*3 floats (out_cap_floats must be at least that).  A payload with
 * fewer floats than its header announces is PT_ERR_PARSE (the reference panics);
 * trailing bytes are ignored as Rgb32FImage::from_vec ignores them. */
PtStatus pt_rendered_image_from_bytes(const uint8_t* bytes, size_t size, uint32_t* out_width,
                                      uint32_t* out_height, float* out_rgb, size_t out_cap_floats);

/* ------------------------------------------------ the worker's front end */

/* worker::discovery::{Request, Response} (worker/src/lib.rs:81-91) in the wire format the
 * reference uses for them, postcard 1.1: the worker answers every datagram on UDP port
 * 40000 that decodes as a Request with a Response naming its WebSocket port
 * (worker/src/main.rs:95-113; the client's side is client/src/worker_pool.rs:137-157).
 * `Request {}` is zero bytes; `Response { websocket_port: u16 }` is the port as an LEB128
 * varint (30000 = B0 EA 01).  Decoders ignore trailing bytes like postcard::from_bytes. */
PtStatus pt_discovery_request_encode(uint8_t* out, size_t cap, size_t* out_len);
PtStatus pt_discovery_request_decode(const uint8_t* bytes, size_t len);
PtStatus pt_discovery_response_encode(uint16_t websocket_port, uint8_t* out, size_t cap, size_t* out_len);
PtStatus pt_discovery_response_decode(const uint8_t* bytes, size_t len, uint16_t* out_websocket_port);

/* The digest the client names a scene's file bucket with (client/src/scene.rs:19-85): MD5
 * of every file the scene refers to, sorted by path, folded into one; 32 lower-case hex
 * digits and a NUL into out_hex33.  kd_tree files are left out (the client stops at them
 * with unimplemented!(), scene.rs:59-61; here they are a cache, not scene content). */
PtStatus pt_scene_md5(const char* root_dir, const char* scene_path, char* out_hex33);

/* Worker (worker/src/lib.rs:18-49): scenes cached by scene_md5, files fetched from the
 * scene's bucket -- here the directory store_root + "/" + scene_md5 instead of the GridFS
 * bucket of that name (file_store.rs:12-35).  A task without scene_md5 (the file form,
 * render_task.rs:5-10) is not a RenderTask and is refused like any frame that does not
 * decode.  The scene name of a frame and the names inside the files it leads to are a
 * peer's: a name that is absolute or has a ".." component is PT_ERR_INVALID_ARGUMENT, so
 * nothing is read or written outside the bucket.  Missing or stale .tree files of a
 * bucket are written back on the first load. */
typedef struct PtWorker PtWorker;
PtStatus pt_worker_create(const char* store_root, int device, PtWorker** out_worker);
void pt_worker_destroy(PtWorker* worker);
uint32_t pt_worker_cached_scene_count(const PtWorker* worker);

/* One round of connection_loop (worker/src/main.rs:68-93) without the socket: task_json is
 * the text frame (a RenderTask), *out_bytes / *out_size the binary frame sent back
 * (RenderedImage::to_bytes, lib.rs:56-62; release with pt_buffer_free).  A frame that does
 * not decode is PT_ERR_PARSE with the reference's "Failed to decode render task" message;
 * the reference then drops the connection.  opts may be NULL. */
PtStatus pt_worker_handle_message(PtWorker* worker, const char* task_json, const PtRenderOpts* opts,
                                  uint8_t** out_bytes, uint64_t* out_size);

/* The worker's two sockets (worker/src/main.rs:36-113).  A TCP listener on websocket_port
 * (main.rs:17: 30000) whose connections are upgraded to WebSockets (RFC 6455; what
 * tokio_tungstenite::accept_async does, main.rs:53-55) and then run connection_loop
 * (main.rs:68-93): one text frame holding a RenderTask in, pt_worker_handle_message, one
 * binary frame holding RenderedImage::to_bytes out, until a round fails -- a frame that is
 * not text ("Unexpected message format"), a task that does not decode or a render that fails
 * ends that connection and no other (main.rs:60-65).  A UDP socket on udp_port (main.rs:18:
 * 40000, SO_BROADCAST) answers every datagram, which is how discovery::Request {} decodes
 * (main.rs:95-113), with the postcard bytes of Response { websocket_port } naming the port
 * that was actually bound.  Port 0 binds any free port; bind_addr NULL is "0.0.0.0"
 * (main.rs:37,96).  Renders of all connections take turns behind the worker's lock
 * (main.rs:32,82).  Differences from the reference, all on the tolerant side: a ping is
 * answered with a pong and the connection goes on (tungstenite hands the ping to
 * connection_loop, which gives up on it), a close frame is answered with a close frame, and
 * a client that disconnects ends its connection thread quietly (main.rs:72 panics the task).
 * The worker must outlive the server. */
typedef struct PtServer PtServer;
typedef struct PtServerStats {
    uint64_t connections;        /* TCP connections accepted */
    uint64_t handshakes;         /* of those, upgraded to a WebSocket */
    uint64_t messages_answered;  /* rounds of connection_loop that sent a binary frame */
    uint64_t messages_failed;    /* rounds that ended their connection */
    uint64_t datagrams_answered; /* discovery requests answered */
} PtServerStats;
PtStatus pt_worker_serve_start(PtWorker* worker, const char* bind_addr, uint16_t websocket_port, uint16_t udp_port,
                               PtServer** out_server);
PtStatus pt_server_ports(const PtServer* server, uint16_t* out_websocket_port, uint16_t* out_udp_port);
/* Counters so far; last_error (may be NULL) receives the message of the round or handshake
 * that failed last ("Error during message exchange: ..." in the reference's log). */
PtStatus pt_server_stats(const PtServer* server, PtServerStats* out_stats, char* last_error, size_t last_error_cap);
/* Closes both sockets, ends the open connections, joins the threads, frees the server. */
void pt_server_stop(PtServer* server);
/* start_ws (main.rs:36-48) as one blocking call: serves until *stop_flag becomes non-zero
 * (read as a volatile ten times a second; NULL = until the process ends). */
PtStatus pt_worker_serve(PtWorker* worker, const char* bind_addr, uint16_t websocket_port, uint16_t udp_port,
                         const int32_t* stop_flag);

/* ------------------------------------------------- the client's frame sum */

/* Frame / RenderSum (client/src/frame.rs:11-15,54-58): the element-wise sum of
 * the renders received for one frame and their number, kept on `device`. */
typedef struct PtFrame PtFrame;

/* Frame::new (frame.rs:22-33): a zeroed width x height sum, count 0. */
PtStatus pt_frame_create(uint32_t width, uint32_t height, int device, PtFrame** out_frame);
void pt_frame_destroy(PtFrame* frame);

/* RenderSum::add_render (frame.rs:61-71): sum += rgb (host memory, width*height*3
 * floats in the layout pt_render returns), count += 1. */
PtStatus pt_frame_add_render(PtFrame* frame, const float* rgb);

/* The same with the render already in device memory of the frame's device (for
 * instance pt_resolve_device's output); enqueued on `stream`, returns at once. */
PtStatus pt_frame_add_render_device(PtFrame* frame, const float* d_rgb, void* stream);

/* self.count (frame.rs:57). */
uint64_t pt_frame_render_count(const PtFrame* frame);

/* RenderSum::into_image (frame.rs:73-83) and gamma_correction (frame.rs:86-103):
 * out_rgb (host, width*height*3 floats, may be NULL) = sum / max(count, 1);
 * out_rgba (host, width*height*4 bytes, may be NULL) = per channel
 * (c / (1 + c) * 255) as u8 of that mean -- truncating, saturating, NaN -> 0 as
 * Rust's cast does -- with alpha 255.  Waits for everything queued on the device. */
PtStatus pt_frame_image(PtFrame* frame, float* out_rgb, uint8_t* out_rgba);

/* The same into device memory (either may be NULL; d_out_rgba 4-byte aligned),
 * enqueued on `stream`. */
PtStatus pt_frame_image_device(PtFrame* frame, float* d_out_rgb, uint8_t* d_out_rgba, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PATHTRACING_AMD_H */
