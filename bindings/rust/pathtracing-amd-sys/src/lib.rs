//! Raw bindings to `libpathtracing_amd.so`: one `#[repr(C)]` struct per struct of
//! `include/pathtracing_amd.h` (same field order, same names) and one `extern "C"`
//! declaration per entry point.  Nothing here allocates or owns anything.
//!
//! The math mirrors are field-for-field the reference's `math` crate types
//! (math/src/lib.rs: Vec2 :118-123, Vec3 :214-220, Mat3 :369-375, Mat4 :544-551),
//! `PtRay` is worker/src/ray.rs:3-9 and `PtCamera` is worker/src/camera.rs:35-48
//! with `usize` widened to `u64` and the enum as a `u32` tag.
//!
//! Source only: the environment this library is built and tested in has no Rust
//! toolchain, so this crate has not been compiled there.  `tests/test_rust_binding_source.py`
//! keeps its structs and functions in step with the C header.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const PT_ABI_VERSION: u32 = 3;
pub const PT_NO_INDEX: u32 = 0xFFFF_FFFF;
pub const PT_FLAG_BRUTE_FORCE: u32 = 1;
pub const PT_FLAG_PROFILE: u32 = 2;
pub const PT_FLAG_WAVEFRONT: u32 = 4;
pub const PT_FLAG_WAVEFRONT_SPLIT: u32 = 8;
pub const PT_FLAG_SINGLE_STREAM: u32 = 16;
pub const PT_FLAG_COUNT_VISITS: u32 = 32;
pub const PT_STATIC_TRACE_MAX_RAYS: u32 = 16777216;

pub const PT_SHARD_SAMPLES: u32 = 0;
pub const PT_SHARD_ROWS: u32 = 1;

/// `PtStatus`
pub type PtStatus = c_int;
pub const PT_OK: PtStatus = 0;
pub const PT_ERR_INVALID_ARGUMENT: PtStatus = 1;
pub const PT_ERR_SCENE: PtStatus = 2;
pub const PT_ERR_IO: PtStatus = 3;
pub const PT_ERR_PARSE: PtStatus = 4;
pub const PT_ERR_DEVICE: PtStatus = 5;
pub const PT_ERR_UNSUPPORTED: PtStatus = 6;

/// `PtNodeType`
pub const PT_NODE_COLLECTION: u32 = 0;
pub const PT_NODE_TRANSFORM: u32 = 1;
pub const PT_NODE_MESH: u32 = 2;
pub const PT_NODE_SPHERE: u32 = 3;
pub const PT_NODE_PLANE: u32 = 4;
pub const PT_NODE_KD_TREE: u32 = 5;

/// `PtMaterialType`, `PtUvMode`, `PtBokehShape`
pub const PT_MATERIAL_BASE: u32 = 0;
pub const PT_MATERIAL_PBR: u32 = 1;
pub const PT_UV_CLAMP: u32 = 0;
pub const PT_UV_REPEAT: u32 = 1;
pub const PT_BOKEH_POINT: u32 = 0;
pub const PT_BOKEH_CIRCLE: u32 = 1;
pub const PT_BOKEH_SQUARE: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PtVec2 {
    pub x: f32,
    pub y: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PtVec3 {
    pub x: f32,
    pub y: f32,
    pub z: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PtMat3 {
    pub row0: PtVec3,
    pub row1: PtVec3,
    pub row2: PtVec3,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PtMat4 {
    pub m: [[f32; 4]; 4],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PtRay {
    pub source: PtVec3,
    pub direction: PtVec3,
    pub min: f32,
    pub max: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtNodeDesc {
    pub r#type: u32,
    pub child_count: u32,
    pub matrix: PtMat4,
    pub mesh_id: u32,
    pub material_id: u32,
    pub center: PtVec3,
    pub radius: f32,
    pub point: PtVec3,
    pub normal: PtVec3,
    pub tangent: PtVec3,
    pub bitangent: PtVec3,
    pub tree_id: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtMeshDesc {
    pub triangle_count: u64,
    pub positions: *const f32,
    pub normals: *const f32,
    pub normal_valid: *const u8,
    pub uvs: *const f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtMaterialDesc {
    pub r#type: u32,
    pub color_is_texture: u32,
    pub color: PtVec3,
    pub image_id: u32,
    pub uv_mode: u32,
    pub emission: PtVec3,
    pub refraction: f32,
    pub reflective: f32,
    pub emissive: f32,
    pub refractive: f32,
    pub albedo: PtVec3,
    pub roughness: f32,
    pub metallic: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtImageDesc {
    pub width: u32,
    pub height: u32,
    pub rgba8: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtTreeBlob {
    pub data: *const u8,
    pub size: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtSceneDesc {
    pub nodes: *const PtNodeDesc,
    pub node_count: u32,
    pub meshes: *const PtMeshDesc,
    pub mesh_count: u32,
    pub materials: *const PtMaterialDesc,
    pub material_count: u32,
    pub images: *const PtImageDesc,
    pub image_count: u32,
    pub trees: *const PtTreeBlob,
    pub tree_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtCamera {
    pub resolution_x: u64,
    pub resolution_y: u64,
    pub rotation: PtMat3,
    pub position: PtVec3,
    pub fov: f32,
    pub near_plane: f32,
    pub focal_length: f32,
    pub bokeh_shape: u32,
    pub bokeh_size: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtRenderTask {
    pub camera: PtCamera,
    pub trace_depth: u64,
    pub iterations: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtRenderOpts {
    pub seed: u64,
    pub sample_offset: u64,
    pub flags: u32,
    pub tile_x0: u32,
    pub tile_y0: u32,
    pub tile_w: u32,
    pub tile_h: u32,
    pub paths_per_batch: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtSceneInfo {
    pub mesh_count: u32,
    pub trees_offered: u32,
    pub trees_adopted: u32,
    pub trees_built: u32,
    pub instance_run_children: u32,
    pub triangle_count: u64,
    pub bvh_bytes: u64,
    pub build_ms: f64,
    pub setup_ms: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtRenderStats {
    pub rays: u64,
    pub paths: u64,
    pub render_ms: f64,
    pub bvh_bytes: u64,
    pub searches: u64,
    pub search_kernel_ms: f64,
    pub search_kernel_launches: u64,
    pub node_visits: u64,
    pub triangle_tests: u64,
    pub wave_node_steps: u64,
    pub wave_leaf_steps: u64,
    pub wave_leaf_rounds: u64,
    pub reduce_ms: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtHit {
    pub hit: u32,
    pub hit_inside: u32,
    pub node_index: u32,
    pub tri_index: u32,
    pub material_id: u32,
    pub t: f32,
    pub point: PtVec3,
    pub normal: PtVec3,
    pub uv: PtVec2,
}

/// Opaque handles.
#[repr(C)]
pub struct PtScene {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PtHostScene {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PtFrame {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PtWorker {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PtServer {
    _private: [u8; 0],
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtServerStats {
    pub connections: u64,
    pub handshakes: u64,
    pub messages_answered: u64,
    pub messages_failed: u64,
    pub datagrams_answered: u64,
}

extern "C" {
    pub fn pt_last_error(buf: *mut c_char, cap: usize) -> usize;
    pub fn pt_abi_version() -> u32;

    pub fn pt_scene_create(desc: *const PtSceneDesc, device: c_int, out_scene: *mut *mut PtScene) -> PtStatus;
    pub fn pt_scene_destroy(scene: *mut PtScene);
    pub fn pt_scene_info(scene: *const PtScene, out_info: *mut PtSceneInfo) -> PtStatus;
    pub fn pt_device_release_workspace(device: c_int) -> PtStatus;

    pub fn pt_render(
        scene: *mut PtScene,
        task: *const PtRenderTask,
        opts: *const PtRenderOpts,
        out_rgb: *mut f32,
        stats: *mut PtRenderStats,
    ) -> PtStatus;
    pub fn pt_render_accumulate_device(
        scene: *mut PtScene,
        task: *const PtRenderTask,
        opts: *const PtRenderOpts,
        d_accum: *mut f32,
        stream: *mut c_void,
        stats: *mut PtRenderStats,
    ) -> PtStatus;
    pub fn pt_resolve_device(
        d_accum: *const f32,
        width: u64,
        height: u64,
        total_iterations: u64,
        d_out_rgb: *mut f32,
        stream: *mut c_void,
    ) -> PtStatus;

    pub fn pt_shard(
        task: *const PtRenderTask,
        opts: *const PtRenderOpts,
        mode: u32,
        world: u32,
        rank: u32,
        out_task: *mut PtRenderTask,
        out_opts: *mut PtRenderOpts,
        out_active: *mut u32,
    ) -> PtStatus;
    pub fn pt_render_multi(
        scenes: *mut *mut PtScene,
        n_devices: c_int,
        task: *const PtRenderTask,
        opts: *const PtRenderOpts,
        mode: u32,
        out_rgb: *mut f32,
        stats: *mut PtRenderStats,
    ) -> PtStatus;
    pub fn pt_trace_closest(
        scene: *mut PtScene,
        rays: *const PtRay,
        n: usize,
        flags: u32,
        out_hits: *mut PtHit,
    ) -> PtStatus;
    pub fn pt_trace_closest_device_static(
        scene: *mut PtScene,
        d_rays: *const PtRay,
        n: usize,
        flags: u32,
        d_hits: *mut PtHit,
        stream: *mut c_void,
    ) -> PtStatus;
    pub fn pt_generate_rays(
        camera: *const PtCamera,
        seed: u64,
        pixel_xy: *const u32,
        sample_index: *const u64,
        n: usize,
        device: c_int,
        out_rays: *mut PtRay,
    ) -> PtStatus;
    pub fn pt_shade(
        scene: *mut PtScene,
        directions: *const PtVec3,
        hits: *const PtHit,
        stream_id: *const u64,
        seed: u64,
        n: usize,
        out_kind: *mut u32,
        out_color_or_multiplier: *mut PtVec3,
        out_direction: *mut PtVec3,
    ) -> PtStatus;

    pub fn pt_host_scene_load(root_dir: *const c_char, scene_path: *const c_char, out: *mut *mut PtHostScene) -> PtStatus;
    pub fn pt_host_scene_desc(scene: *const PtHostScene) -> *const PtSceneDesc;
    pub fn pt_host_scene_destroy(scene: *mut PtHostScene);
    pub fn pt_host_scene_cache_trees(scene: *mut PtHostScene, root_dir: *const c_char, out_written: *mut u32) -> PtStatus;
    pub fn pt_render_task_parse(
        json_text: *const c_char,
        out_task: *mut PtRenderTask,
        scene_path_buf: *mut c_char,
        scene_path_cap: usize,
    ) -> PtStatus;

    pub fn pt_tree_build(mesh: *const PtMeshDesc, out_data: *mut *mut u8, out_size: *mut u64) -> PtStatus;
    pub fn pt_buffer_free(data: *mut u8);
    pub fn pt_tree_validate(mesh: *const PtMeshDesc, data: *const u8, size: u64) -> PtStatus;

    pub fn pt_rendered_image_to_bytes(rgb: *const f32, width: u32, height: u32, out: *mut u8, out_cap: usize) -> PtStatus;
    pub fn pt_rendered_image_from_bytes(
        bytes: *const u8,
        size: usize,
        out_width: *mut u32,
        out_height: *mut u32,
        out_rgb: *mut f32,
        out_cap_floats: usize,
    ) -> PtStatus;

    pub fn pt_discovery_request_encode(out: *mut u8, cap: usize, out_len: *mut usize) -> PtStatus;
    pub fn pt_discovery_request_decode(bytes: *const u8, len: usize) -> PtStatus;
    pub fn pt_discovery_response_encode(websocket_port: u16, out: *mut u8, cap: usize, out_len: *mut usize) -> PtStatus;
    pub fn pt_discovery_response_decode(bytes: *const u8, len: usize, out_websocket_port: *mut u16) -> PtStatus;
    pub fn pt_scene_md5(root_dir: *const c_char, scene_path: *const c_char, out_hex33: *mut c_char) -> PtStatus;
    pub fn pt_worker_create(store_root: *const c_char, device: c_int, out_worker: *mut *mut PtWorker) -> PtStatus;
    pub fn pt_worker_destroy(worker: *mut PtWorker);
    pub fn pt_worker_cached_scene_count(worker: *const PtWorker) -> u32;
    pub fn pt_worker_handle_message(
        worker: *mut PtWorker,
        task_json: *const c_char,
        opts: *const PtRenderOpts,
        out_bytes: *mut *mut u8,
        out_size: *mut u64,
    ) -> PtStatus;
    pub fn pt_worker_serve_start(
        worker: *mut PtWorker,
        bind_addr: *const c_char,
        websocket_port: u16,
        udp_port: u16,
        out_server: *mut *mut PtServer,
    ) -> PtStatus;
    pub fn pt_server_ports(server: *const PtServer, out_websocket_port: *mut u16, out_udp_port: *mut u16) -> PtStatus;
    pub fn pt_server_stats(
        server: *const PtServer,
        out_stats: *mut PtServerStats,
        last_error: *mut c_char,
        last_error_cap: usize,
    ) -> PtStatus;
    pub fn pt_server_stop(server: *mut PtServer);
    pub fn pt_worker_serve(
        worker: *mut PtWorker,
        bind_addr: *const c_char,
        websocket_port: u16,
        udp_port: u16,
        stop_flag: *const i32,
    ) -> PtStatus;
    pub fn pt_frame_create(width: u32, height: u32, device: c_int, out_frame: *mut *mut PtFrame) -> PtStatus;
    pub fn pt_frame_destroy(frame: *mut PtFrame);
    pub fn pt_frame_add_render(frame: *mut PtFrame, rgb: *const f32) -> PtStatus;
    pub fn pt_frame_add_render_device(frame: *mut PtFrame, d_rgb: *const f32, stream: *mut c_void) -> PtStatus;
    pub fn pt_frame_render_count(frame: *const PtFrame) -> u64;
    pub fn pt_frame_image(frame: *mut PtFrame, out_rgb: *mut f32, out_rgba: *mut u8) -> PtStatus;
    pub fn pt_frame_image_device(
        frame: *mut PtFrame,
        d_out_rgb: *mut f32,
        d_out_rgba: *mut u8,
        stream: *mut c_void,
    ) -> PtStatus;
}

/// The calling thread's last error message (what replaces the reference's panics).
pub fn last_error() -> String {
    let mut buf = vec![0u8; 1024];
    let n = unsafe { pt_last_error(buf.as_mut_ptr() as *mut c_char, buf.len()) };
    buf.truncate(n.min(buf.len() - 1));
    String::from_utf8_lossy(&buf).into_owned()
}
