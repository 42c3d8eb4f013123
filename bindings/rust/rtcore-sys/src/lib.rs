//! Raw declarations of `include/rtcore.h` (librtcore.so).  One item per item of
//! the header, same names, same order; see the header for what each entry point
//! stands in for in the reference.
#![allow(non_camel_case_types)]
#![no_std]

use core::ffi::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = 1;
pub const RT_ERR_HIP: c_int = 2;
pub const RT_ERR_NOMEM: c_int = 3;
pub const RT_ERR_LIMIT: c_int = 4;
pub const RT_NO_HIT: u32 = 0xFFFF_FFFF;
pub const RT_RENDER_VISIT_STATS: u32 = 1;
pub const RT_SHARD_SAMPLE: c_int = 0;
pub const RT_SHARD_TILE: c_int = 1;
pub const RT_LDS_STACK_ENTRIES: u32 = 24;
pub const RT_STAT_DEPTHS: usize = 8;
pub const RT_NOALLOC_MAX_RAYS: u32 = 268_435_456;
pub const RT_NOALLOC_MAX_STACK: u32 = 56;
pub const RT_NOALLOC_SCRATCH_BYTES: u32 = 17_825_792;

#[repr(C)]
pub struct rt_ctx {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct rt_scene {
    _opaque: [u8; 0],
}

/// `StreamRay` of src/ray.rs:6-12: identical layout, a `&mut [StreamRay]` can be passed as is.
#[repr(C, align(16))]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_stream_ray {
    pub origin_far: [f32; 4],
    pub dir_recip_near: [f32; 4],
    pub direction: [f32; 4],
}

/// `TraceStats` of src/trace_stats.rs:4-12.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct rt_trace_stats {
    pub total_rays: u64,
    pub inner_visit: u64,
    pub inner_rays: u64,
    pub leaf_visit: u64,
    pub leaf_rays: u64,
    pub leaf_objs: u64,
    pub obj_intersect: u64,
}

/// `BvhStats` of src/bvh/build.rs:5-11.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct rt_bvh_stats {
    pub num_leaves: u32,
    pub num_objs: u32,
    pub max_depth: u32,
    pub max_objs: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_scene_info {
    pub num_triangles: u32,
    pub num_inner_nodes: u32,
    pub root_ref: u32,
    pub root_count: u32,
    pub device_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_flat_node {
    pub bx: [f32; 4],
    pub by: [f32; 4],
    pub bz: [f32; 4],
    pub left: u32,
    pub right: u32,
    pub left_count: u32,
    pub right_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_flat_tri {
    pub v0: [f32; 3],
    pub v0v1: [f32; 3],
    pub v0v2: [f32; 3],
    pub id: u32,
    pub pad: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_camera {
    pub eye: [f32; 3],
    pub target: [f32; 3],
    pub up: [f32; 3],
    pub fov_deg: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_light {
    pub pos: [f32; 3],
    pub intensity: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_params {
    pub width: u32,
    pub height: u32,
    pub samples: u32,
    pub sample_offset: u32,
    pub max_bounces: u32,
    pub tile_size: u32,
    pub shard_index: u32,
    pub shard_count: u32,
    pub samples_per_pass: u32,
    pub flags: u32,
    pub seed: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_wide_node {
    pub lo: [[f32; 4]; 3],
    pub hi: [[f32; 4]; 3],
    pub child: [u32; 4],
    pub bin: u32,
    pub pad: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_qwide_node {
    pub origin: [f32; 3],
    pub exps: u32,
    pub base4: u32,
    pub meta: u32,
    pub lo: [u32; 3],
    pub hi: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct rt_shard_plan {
    pub samples: u32,
    pub sample_offset: u32,
    pub shard_index: u32,
    pub shard_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct rt_depth_stats {
    pub rays: [u64; RT_STAT_DEPTHS],
    pub inner_rays: [u64; RT_STAT_DEPTHS],
    pub obj_intersect: [u64; RT_STAT_DEPTHS],
}

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_version() -> *const c_char;

    pub fn rt_ctx_create(device_id: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_ctx_destroy(ctx: *mut rt_ctx);
    pub fn rt_ctx_set_option(ctx: *mut rt_ctx, name: *const c_char, value: i64) -> c_int;
    pub fn rt_ctx_set_max_paths(ctx: *mut rt_ctx, max_paths: u64) -> c_int;

    pub fn rt_scene_create(
        ctx: *mut rt_ctx,
        positions: *const f32,
        num_vertices: u32,
        indices: *const u32,
        num_triangles: u32,
        geom_normals: *const f32,
        out: *mut *mut rt_scene,
    ) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_stats(scene: *const rt_scene, out: *mut rt_bvh_stats) -> c_int;
    pub fn rt_scene_get_info(scene: *const rt_scene, out: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_get_flat(scene: *const rt_scene, nodes: *mut rt_flat_node, tris: *mut rt_flat_tri) -> c_int;

    pub fn rt_bvh_build_host(
        positions: *const f32,
        num_vertices: u32,
        indices: *const u32,
        num_triangles: u32,
        threads: u32,
        nodes: *mut rt_flat_node,
        order: *mut u32,
        info: *mut rt_scene_info,
        stats: *mut rt_bvh_stats,
    ) -> c_int;
    pub fn rt_bvh_build_search_host(
        positions: *const f32,
        num_vertices: u32,
        indices: *const u32,
        num_triangles: u32,
        threads: u32,
        buckets: u32,
        max_leaf: u32,
        nodes: *mut rt_flat_node,
        order: *mut u32,
        info: *mut rt_scene_info,
        stats: *mut rt_bvh_stats,
    ) -> c_int;
    pub fn rt_bvh_wide_host(
        nodes: *const rt_flat_node,
        num_nodes: u32,
        root_ref: u32,
        root_count: u32,
        num_triangles: u32,
        out: *mut rt_wide_node,
        qout: *mut rt_qwide_node,
        node_unit: *mut u32,
        slot_unit: *mut u32,
        num_wide: *mut u32,
        wide_root_ref: *mut u32,
        wide_depth: *mut u32,
        num_units: *mut u32,
    ) -> c_int;
    pub fn rt_bvh_refine_host(
        positions: *const f32,
        num_vertices: u32,
        indices: *const u32,
        num_triangles: u32,
        nodes: *const rt_flat_node,
        num_nodes: u32,
        root_ref: u32,
        root_count: u32,
        order: *const u32,
        max_tris: u32,
        xnodes: *mut rt_flat_node,
        xorder: *mut u32,
        num_xnodes: *mut u32,
        xroot_ref: *mut u32,
        xroot_count: *mut u32,
        extra_depth: *mut u32,
    ) -> c_int;

    pub fn rt_intersect_stream(
        scene: *mut rt_scene,
        rays: *mut rt_stream_ray,
        hits: *mut u32,
        n: u32,
        stats: *mut rt_trace_stats,
    ) -> c_int;
    pub fn rt_intersect_stream_device(
        scene: *mut rt_scene,
        d_rays: *mut rt_stream_ray,
        d_hits: *mut u32,
        n: u32,
        stats: *mut rt_trace_stats,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn rt_intersect_stream_noalloc(
        scene: *mut rt_scene,
        d_rays: *mut rt_stream_ray,
        d_hits: *mut u32,
        n: u32,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn rt_occluded_stream(
        scene: *mut rt_scene,
        rays: *const rt_stream_ray,
        occluded: *mut u8,
        n: u32,
        stats: *mut rt_trace_stats,
    ) -> c_int;
    pub fn rt_intersect1(
        scene: *mut rt_scene,
        ray: *mut rt_stream_ray,
        hit: *mut u32,
        stats: *mut rt_trace_stats,
    ) -> c_int;
    pub fn rt_occluded1(
        scene: *mut rt_scene,
        ray: *const rt_stream_ray,
        occluded: *mut u8,
        stats: *mut rt_trace_stats,
    ) -> c_int;
    pub fn rt_occluded_stream_device(
        scene: *mut rt_scene,
        d_rays: *const rt_stream_ray,
        d_occluded: *mut u8,
        n: u32,
        stats: *mut rt_trace_stats,
        hip_stream: *mut c_void,
    ) -> c_int;

    pub fn rt_render(
        scene: *mut rt_scene,
        camera: *const rt_camera,
        light: *const rt_light,
        params: *const rt_render_params,
        accum: *mut f32,
        stats: *mut rt_trace_stats,
    ) -> c_int;
    pub fn rt_render_device(
        scene: *mut rt_scene,
        camera: *const rt_camera,
        light: *const rt_light,
        params: *const rt_render_params,
        d_accum: *mut f32,
        stats: *mut rt_trace_stats,
        hip_stream: *mut c_void,
    ) -> c_int;

    pub fn rt_shard_plan_make(
        rank: u32,
        world: u32,
        mode: c_int,
        samples: u32,
        sample_offset: u32,
        out: *mut rt_shard_plan,
    ) -> c_int;
    pub fn rt_node_render(
        scene: *mut rt_scene,
        camera: *const rt_camera,
        light: *const rt_light,
        frame: *const rt_render_params,
        rank: u32,
        world: u32,
        mode: c_int,
        nccl_comm: *mut c_void,
        d_accum: *mut f32,
        stats: *mut rt_trace_stats,
        hip_stream: *mut c_void,
    ) -> c_int;

    pub fn rt_generate_primary(
        ctx: *mut rt_ctx,
        camera: *const rt_camera,
        params: *const rt_render_params,
        rays: *mut rt_stream_ray,
    ) -> c_int;
    pub fn rt_resolve(ctx: *mut rt_ctx, accum: *const f32, width: u32, height: u32, rgb: *mut u8) -> c_int;

    pub fn rt_ctx_last_kernel_ms(ctx: *mut rt_ctx, ms: *mut f32, launches: *mut u32) -> c_int;
    pub fn rt_ctx_last_launch_ms(ctx: *mut rt_ctx, ms: *mut f32, kind: *mut u32, cap: u32, count: *mut u32) -> c_int;
    pub fn rt_debug_sort_keys(ctx: *mut rt_ctx, keys: *const u32, n: u32, bits: u32, perm: *mut u32) -> c_int;
    pub fn rt_ctx_last_stats_split(ctx: *mut rt_ctx, out: *mut rt_trace_stats) -> c_int;
    pub fn rt_ctx_last_max_ray_steps(ctx: *mut rt_ctx, out: *mut u64) -> c_int;
    pub fn rt_ctx_last_depth_stats(ctx: *mut rt_ctx, out: *mut rt_depth_stats) -> c_int;
    pub fn rt_ctx_last_max_stack_entries(ctx: *mut rt_ctx, out: *mut u64) -> c_int;
    pub fn rt_ctx_last_pass_paths(ctx: *mut rt_ctx, out: *mut u64) -> c_int;
    pub fn rt_bvh_build_search_device(
        ctx: *mut rt_ctx,
        positions: *const f32,
        num_vertices: u32,
        indices: *const u32,
        num_triangles: u32,
        buckets: u32,
        max_leaf: u32,
        nodes: *mut rt_flat_node,
        order: *mut u32,
        info: *mut rt_scene_info,
        stats: *mut rt_bvh_stats,
    ) -> c_int;
    pub fn rt_scene_fast_tree_on_device(scene: *const rt_scene, out: *mut c_int) -> c_int;
    pub fn rt_scene_get_wide_items(
        scene: *const rt_scene,
        units: *mut c_void,
        capacity_units: u64,
        num_units: *mut u64,
    ) -> c_int;
    pub fn rt_scene_bounds(scene: *const rt_scene, lo: *mut f32, hi: *mut f32) -> c_int;
}

// Layout checks against the sizes include/rtcore.h implies.
const _: () = {
    assert!(core::mem::size_of::<rt_stream_ray>() == 48);
    assert!(core::mem::size_of::<rt_trace_stats>() == 56);
    assert!(core::mem::size_of::<rt_flat_node>() == 64);
    assert!(core::mem::size_of::<rt_flat_tri>() == 48);
    assert!(core::mem::size_of::<rt_wide_node>() == 128);
    assert!(core::mem::size_of::<rt_qwide_node>() == 64);
    assert!(core::mem::size_of::<rt_render_params>() == 48);
    assert!(core::mem::size_of::<rt_camera>() == 40);
    assert!(core::mem::size_of::<rt_light>() == 24);
};
