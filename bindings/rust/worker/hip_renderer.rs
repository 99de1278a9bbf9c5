//! `worker/src/renderer/hip_renderer.rs` -- the second implementor of the worker's
//! `Renderer` trait (worker/src/renderer/mod.rs:9-13), next to `CPURenderer`
//! (cpu_renderer/mod.rs:178-195), rendering on an MI355X through
//! `libpathtracing_amd.so`.
//!
//! Source only: the environment this library is built and tested in has no Rust
//! toolchain, so this file has not been compiled there.
//!
//! What the reference needs besides this file
//! ------------------------------------------
//! * `worker/Cargo.toml`: `pathtracing-amd-sys = { path = ".../bindings/rust/pathtracing-amd-sys" }`
//! * `worker/src/renderer/mod.rs`: `pub mod hip_renderer;`
//! * `worker/src/lib.rs:44`: `let mut renderer = HipRenderer::init(scene);` -- and, because a
//!   renderer is constructed per task there while the scene is cached per `scene_md5`
//!   (lib.rs:32-43), keep the `HipRenderer` beside the cached `Arc<Scene>`: creating it builds the
//!   acceleration structures and uploads the scene.
//! * The scene graph only exposes `trace_ray` (cpu_renderer/mod.rs:41-43) and the materials only
//!   `get_color` (:50-57), so each gets one more method, implemented in the type's own module
//!   (the fields are private); the bodies are listed at the end of this file:
//!       trait SceneNode { ...; fn flatten(&self, out: &mut Vec<PtNodeDesc>); }
//!       trait Material  { ...; fn describe(&self) -> PtMaterialDesc; }
//!   plus `Mesh::triangles(&self) -> &[Triangle]`, `Triangle::vertices(&self) -> &[Vertex; 3]`
//!   and `Image::as_raw(&self) -> &[u8]` accessors.

use std::sync::Arc;

use image::Rgb32FImage;
use pathtracing_amd_sys::*;

use super::Renderer;
use crate::{api::render_task::RenderTask, camera::BokehShape, scene::Scene};

pub struct HipRenderer {
    scene: *mut PtScene,
}

// The handle is used by one render at a time: the worker sits behind a
// tokio::sync::Mutex that is held across the render (main.rs:32, :82).
unsafe impl Send for HipRenderer {}

/// Host arrays the description points into; dropped after `pt_scene_create`, which copies everything.
struct FlatScene {
    nodes: Vec<PtNodeDesc>,
    positions: Vec<Vec<f32>>,
    normals: Vec<Vec<f32>>,
    uvs: Vec<Vec<f32>>,
    meshes: Vec<PtMeshDesc>,
    materials: Vec<PtMaterialDesc>,
    images: Vec<PtImageDesc>,
}

fn flatten(scene: &Scene) -> FlatScene {
    let mut flat = FlatScene {
        nodes: vec![],
        positions: vec![],
        normals: vec![],
        uvs: vec![],
        meshes: vec![],
        materials: vec![],
        images: vec![],
    };
    // depth first, every node before its children, children in Vec order: the order
    // NodeCollection::trace_ray visits them in (node_collection.rs:43-55) decides ties
    scene.hierarchy.flatten(&mut flat.nodes);
    for mesh in &scene.meshes {
        // triangles in Vec order: the order Mesh::trace_ray scans them in (mesh/mod.rs:57-62) decides ties.
        // The vertex normals are the initialised ones (face normal where the file gave none,
        // triangle/mod.rs:65-84), so every triangle's normals are valid.
        let n = mesh.triangles().len();
        let (mut p, mut nr, mut uv) = (Vec::with_capacity(9 * n), Vec::with_capacity(9 * n), Vec::with_capacity(6 * n));
        for triangle in mesh.triangles() {
            for v in triangle.vertices() {
                p.extend_from_slice(&[v.position.x, v.position.y, v.position.z]);
                nr.extend_from_slice(&[v.normal.x, v.normal.y, v.normal.z]);
                uv.extend_from_slice(&[v.uv.x, v.uv.y]);
            }
        }
        flat.positions.push(p);
        flat.normals.push(nr);
        flat.uvs.push(uv);
    }
    for i in 0..scene.meshes.len() {
        flat.meshes.push(PtMeshDesc {
            triangle_count: (flat.positions[i].len() / 9) as u64,
            positions: flat.positions[i].as_ptr(),
            normals: flat.normals[i].as_ptr(),
            normal_valid: std::ptr::null(),
            uvs: flat.uvs[i].as_ptr(),
        });
    }
    flat.materials = scene.materials.iter().map(|m| m.describe()).collect();
    flat.images = scene
        .images
        .iter()
        .map(|i| PtImageDesc { width: i.width(), height: i.height(), rgba8: i.as_raw().as_ptr() })
        .collect();
    flat
}

#[async_trait::async_trait]
impl Renderer for HipRenderer {
    fn init(scene: Arc<Scene>) -> HipRenderer {
        let flat = flatten(&scene);
        let desc = PtSceneDesc {
            nodes: flat.nodes.as_ptr(),
            node_count: flat.nodes.len() as u32,
            meshes: flat.meshes.as_ptr(),
            mesh_count: flat.meshes.len() as u32,
            materials: flat.materials.as_ptr(),
            material_count: flat.materials.len() as u32,
            images: flat.images.as_ptr(),
            image_count: flat.images.len() as u32,
            trees: std::ptr::null(), // kd_tree nodes carry PT_NO_INDEX: the library builds
            tree_count: 0,
        };
        let mut handle = std::ptr::null_mut();
        let rc = unsafe { pt_scene_create(&desc, 0, &mut handle) };
        assert_eq!(rc, PT_OK, "{}", last_error()); // the reference panics on every failure as well
        HipRenderer { scene: handle }
    }

    async fn render(&mut self, render_task: Arc<RenderTask>) -> Rgb32FImage {
        let camera = &render_task.camera;
        let (w, h) = (camera.resolution.x, camera.resolution.y);
        let (bokeh_shape, bokeh_size) = match camera.bokeh_shape {
            BokehShape::Point => (PT_BOKEH_POINT, camera.bokeh_size),
            BokehShape::Circle => (PT_BOKEH_CIRCLE, camera.bokeh_size),
            BokehShape::Square => (PT_BOKEH_SQUARE, camera.bokeh_size),
        };
        let v3 = |v: math::Vec3| PtVec3 { x: v.x, y: v.y, z: v.z };
        let task = PtRenderTask {
            camera: PtCamera {
                resolution_x: w as u64,
                resolution_y: h as u64,
                rotation: PtMat3 { row0: v3(camera.rotation.row0), row1: v3(camera.rotation.row1), row2: v3(camera.rotation.row2) },
                position: v3(camera.position),
                fov: camera.fov,
                near_plane: camera.near_plane,
                focal_length: camera.focal_length,
                bokeh_shape,
                bokeh_size,
            },
            trace_depth: render_task.config.trace_depth as u64,
            iterations: render_task.config.iterations as u64,
        };
        // The reference's random numbers are unseeded (thread_rng); the library is deterministic,
        // so a worker that wants fresh noise per task supplies a seed of its own.
        let opts = PtRenderOpts { seed: rand::random(), ..Default::default() };
        let mut buf = vec![0f32; w * h * 3];
        let rc = unsafe { pt_render(self.scene, &task, &opts, buf.as_mut_ptr(), std::ptr::null_mut()) };
        assert_eq!(rc, PT_OK, "{}", last_error());
        // rows already flipped like get_image (cpu_renderer/mod.rs:143-175)
        Rgb32FImage::from_raw(w as u32, h as u32, buf).unwrap()
    }
}

impl Drop for HipRenderer {
    fn drop(&mut self) {
        unsafe { pt_scene_destroy(self.scene) }
    }
}

// ---------------------------------------------------------------------------------------------
// The `flatten` / `describe` bodies, each for the module that owns the type.
//
// scene/hierarchy/node_collection.rs
//     fn flatten(&self, out: &mut Vec<PtNodeDesc>) {
//         out.push(PtNodeDesc { r#type: PT_NODE_COLLECTION, child_count: self.children.len() as u32, ..Default::default() });
//         for child in &self.children { child.flatten(out); }
//     }
// scene/hierarchy/transform.rs      (matrix_inverse / normal_matrix are recomputed by the library
//                                    with the arithmetic of math lib.rs:553-649)
//     fn flatten(&self, out: &mut Vec<PtNodeDesc>) {
//         let row = |r: &math::Vec4| [r.x, r.y, r.z, r.w];
//         let m = [row(&self.matrix.row0), row(&self.matrix.row1), row(&self.matrix.row2), row(&self.matrix.row3)];
//         out.push(PtNodeDesc { r#type: PT_NODE_TRANSFORM, child_count: 1, matrix: PtMat4 { m }, ..Default::default() });
//         self.child.flatten(out);
//     }
// scene/hierarchy/kd_tree.rs
//     fn flatten(&self, out: &mut Vec<PtNodeDesc>) {
//         out.push(PtNodeDesc { r#type: PT_NODE_KD_TREE, child_count: 1, tree_id: PT_NO_INDEX, ..Default::default() });
//         self.child.flatten(out);
//     }
// scene/hierarchy/mesh.rs
//     fn flatten(&self, out: &mut Vec<PtNodeDesc>) {
//         out.push(PtNodeDesc { r#type: PT_NODE_MESH, mesh_id: self.path as u32, material_id: self.material as u32, ..Default::default() });
//     }
// scene/hierarchy/sphere.rs
//     fn flatten(&self, out: &mut Vec<PtNodeDesc>) {
//         out.push(PtNodeDesc { r#type: PT_NODE_SPHERE, material_id: self.material as u32, center: v3(self.center),
//                               radius: self.radius, ..Default::default() });
//     }
// scene/hierarchy/plane.rs          (the library normalises the axes as init does, plane.rs:45-47.  Handing it
//                                    the already normalised ones can move their last bit, since normalising is not
//                                    idempotent in f32: for bit-equality with CPURenderer keep the vectors as
//                                    deserialised in the Plane and hand those over)
//     fn flatten(&self, out: &mut Vec<PtNodeDesc>) {
//         out.push(PtNodeDesc { r#type: PT_NODE_PLANE, material_id: self.material as u32, point: v3(self.point),
//                               normal: v3(self.normal), tangent: v3(self.tangent), bitangent: v3(self.bitangent),
//                               ..Default::default() });
//     }
// scene/resource/material/base.rs
//     fn describe(&self) -> PtMaterialDesc {
//         let (color_is_texture, color, image_id, uv_mode) = match &self.color {
//             MaterialInput::Color { color } => (0, v3(*color), 0, PT_UV_CLAMP),
//             MaterialInput::Texture(t) => (1, PtVec3::default(), t.image as u32,
//                                           match t.uv_mode { UvMode::Clamp => PT_UV_CLAMP, UvMode::Repeat => PT_UV_REPEAT }),
//         };
//         PtMaterialDesc { r#type: PT_MATERIAL_BASE, color_is_texture, color, image_id, uv_mode, emission: v3(self.emission),
//                          refraction: self.refraction, reflective: self.reflective, emissive: self.emissive,
//                          refractive: self.refractive, ..Default::default() }
//     }
// scene/resource/material/pbr.rs    (f0 and roughness_sqr are recomputed by the library as pbr.rs:92-96 does)
//     fn describe(&self) -> PtMaterialDesc {
//         PtMaterialDesc { r#type: PT_MATERIAL_PBR, albedo: v3(self.albedo), roughness: self.roughness,
//                          metallic: self.metallic, ..Default::default() }
//     }
// with   fn v3(v: math::Vec3) -> PtVec3 { PtVec3 { x: v.x, y: v.y, z: v.z } }
