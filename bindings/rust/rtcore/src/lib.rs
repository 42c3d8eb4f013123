//! Safe wrapper over `rtcore-sys`.
//!
//! The shapes follow the reference's own call sites so that wiring it in is a
//! few lines: `Scene::new` takes what `TriangleMesh` holds (src/mesh.rs:9-15),
//! `intersect_stream` / `occluded_stream` take the slices
//! `Bvh::intersect_stream` / `Bvh::occluded_stream` take
//! (src/bvh/intersect_stream.rs:10-15, 200-205), `render` replaces the tile
//! loop of src/main.rs:133-161 for a whole frame.
use std::ffi::{CStr, CString};
use std::fmt;
use std::ptr;

pub use rtcore_sys as sys;
pub use rtcore_sys::{
    rt_bvh_stats as BvhStats, rt_camera as Camera, rt_light as Light, rt_render_params as RenderParams,
    rt_stream_ray as StreamRay, rt_trace_stats as TraceStats,
};

/// A failed call: the `RT_ERR_*` code and the library's message for it.
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "rtcore error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

pub type Result<T> = std::result::Result<T, Error>;

fn check(code: i32) -> Result<()> {
    if code == sys::RT_OK {
        return Ok(());
    }
    // rt_last_error() is thread local and valid until this thread's next call
    let message = unsafe { CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned();
    Err(Error { code, message })
}

/// One GPU.  The library allows one in-flight call per context, which `&mut`
/// on the calls that launch work enforces.
pub struct Context {
    raw: *mut sys::rt_ctx,
}
unsafe impl Send for Context {}

impl Context {
    pub fn new(device_id: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::rt_ctx_create(device_id, &mut raw) })?;
        Ok(Self { raw })
    }

    /// Tuning knob by name (`"coop_steps"`, `"traversal_deep"`, ...; see the header).
    pub fn set_option(&mut self, name: &str, value: i64) -> Result<()> {
        let c = CString::new(name).map_err(|_| Error { code: sys::RT_ERR_INVALID, message: "NUL in name".into() })?;
        check(unsafe { sys::rt_ctx_set_option(self.raw, c.as_ptr(), value) })
    }

    /// The reference's exact per-ray walk for every launch (ids, distances and
    /// visit counters equal to `Bvh::intersect_stream`); frames render at 60-75 % of the
    /// default configuration's rate.
    pub fn set_exact(&mut self) -> Result<()> {
        self.set_option("traversal_depth0", 0)?;
        self.set_option("traversal_deep", 1)?;
        self.set_option("traversal_stream", 1)?;
        self.set_option("coop_steps", 0)
    }

    pub fn set_max_paths(&mut self, max_paths: u64) -> Result<()> {
        check(unsafe { sys::rt_ctx_set_max_paths(self.raw, max_paths) })
    }

    /// `color_vec_to_rgb_norm_gamma` (src/main.rs:34-41) over a whole frame.
    pub fn resolve(&mut self, accum: &[[f32; 4]], width: u32, height: u32) -> Result<Vec<u8>> {
        let px = width as usize * height as usize;
        if accum.len() != px {
            return Err(Error { code: sys::RT_ERR_INVALID, message: "accum length != width*height".into() });
        }
        let mut rgb = vec![0u8; 3 * px];
        check(unsafe { sys::rt_resolve(self.raw, accum.as_ptr() as *const f32, width, height, rgb.as_mut_ptr()) })?;
        Ok(rgb)
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::rt_ctx_destroy(self.raw) }
    }
}

/// Device-resident tree and triangles; immutable after creation.  Borrows its
/// context mutably for its lifetime: one in-flight call per context.
pub struct Scene<'c> {
    raw: *mut sys::rt_scene,
    _ctx: &'c mut Context,
}

impl<'c> Scene<'c> {
    /// Builds the reference's binned-SAH tree (src/bvh/build.rs:45-207) over an
    /// indexed mesh and uploads it.  `geom_normals` has one normal per triangle
    /// (src/mesh.rs:180-188).
    pub fn new(ctx: &'c mut Context, positions: &[[f32; 3]], indices: &[[u32; 3]], geom_normals: &[[f32; 3]]) -> Result<Self> {
        if geom_normals.len() != indices.len() {
            return Err(Error { code: sys::RT_ERR_INVALID, message: "one geometric normal per triangle".into() });
        }
        let mut raw = ptr::null_mut();
        check(unsafe {
            sys::rt_scene_create(
                ctx.raw,
                positions.as_ptr() as *const f32,
                positions.len() as u32,
                indices.as_ptr() as *const u32,
                indices.len() as u32,
                geom_normals.as_ptr() as *const f32,
                &mut raw,
            )
        })?;
        Ok(Self { raw, _ctx: ctx })
    }

    /// What `Bvh::build` prints (src/bvh/build.rs:51).
    pub fn bvh_stats(&self) -> Result<BvhStats> {
        let mut s = BvhStats::default();
        check(unsafe { sys::rt_scene_stats(self.raw, &mut s) })?;
        Ok(s)
    }

    /// `Bvh::intersect_stream`: `hits[i]` = mesh triangle id of the closest hit,
    /// `rays[i].origin_far[3]` shrunk to its distance.
    pub fn intersect_stream(&mut self, rays: &mut [StreamRay], hits: &mut [Option<u32>], stats: &mut TraceStats) -> Result<()> {
        assert_eq!(rays.len(), hits.len());
        assert!(rays.len() <= u32::MAX as usize); // src/bvh/intersect_stream.rs:16
        let mut raw = vec![sys::RT_NO_HIT; rays.len()];
        let mut st = TraceStats::default();
        check(unsafe { sys::rt_intersect_stream(self.raw, rays.as_mut_ptr(), raw.as_mut_ptr(), rays.len() as u32, &mut st) })?;
        for (h, r) in hits.iter_mut().zip(raw) {
            *h = if r == sys::RT_NO_HIT { None } else { Some(r) };
        }
        add_stats(stats, &st);
        Ok(())
    }

    /// `Bvh::occluded_stream`: sets `occluded[i]` for rays with any hit in [near, far).
    pub fn occluded_stream(&mut self, rays: &[StreamRay], occluded: &mut [bool], stats: &mut TraceStats) -> Result<()> {
        assert_eq!(rays.len(), occluded.len());
        let mut raw = vec![0u8; rays.len()];
        let mut st = TraceStats::default();
        check(unsafe { sys::rt_occluded_stream(self.raw, rays.as_ptr(), raw.as_mut_ptr(), rays.len() as u32, &mut st) })?;
        for (o, r) in occluded.iter_mut().zip(raw) {
            *o |= r != 0; // the reference only ever sets the flag (src/bvh/intersect_stream.rs:463)
        }
        add_stats(stats, &st);
        Ok(())
    }

    /// `Bvh::intersect1` (src/bvh/intersect1.rs:109-210): closest hit of one ray, in that
    /// function's visit order; `ray.origin_far[3]` is shrunk to the hit distance.
    pub fn intersect1(&mut self, ray: &mut StreamRay, stats: &mut TraceStats) -> Result<Option<u32>> {
        let mut hit = sys::RT_NO_HIT;
        let mut st = TraceStats::default();
        check(unsafe { sys::rt_intersect1(self.raw, ray, &mut hit, &mut st) })?;
        add_stats(stats, &st);
        Ok(if hit == sys::RT_NO_HIT { None } else { Some(hit) })
    }

    /// `Bvh::occluded1` (src/bvh/intersect1.rs:7-106): any hit of one ray in [near, far).
    pub fn occluded1(&mut self, ray: &StreamRay, stats: &mut TraceStats) -> Result<bool> {
        let mut occ = 0u8;
        let mut st = TraceStats::default();
        check(unsafe { sys::rt_occluded1(self.raw, ray, &mut occ, &mut st) })?;
        add_stats(stats, &st);
        Ok(occ != 0)
    }

    /// A whole frame (or one progressive pass: advance `params.sample_offset`)
    /// of the reference's estimator, added into `accum` (sum r, g, b, sample count per pixel).
    pub fn render(&mut self, camera: &Camera, light: &Light, params: &RenderParams, accum: &mut [[f32; 4]], stats: &mut TraceStats) -> Result<()> {
        if accum.len() != params.width as usize * params.height as usize {
            return Err(Error { code: sys::RT_ERR_INVALID, message: "accum length != width*height".into() });
        }
        let mut st = TraceStats::default();
        check(unsafe { sys::rt_render(self.raw, camera, light, params, accum.as_mut_ptr() as *mut f32, &mut st) })?;
        add_stats(stats, &st);
        Ok(())
    }
}

impl Drop for Scene<'_> {
    fn drop(&mut self) {
        unsafe { sys::rt_scene_destroy(self.raw) }
    }
}

fn add_stats(acc: &mut TraceStats, s: &TraceStats) {
    acc.total_rays += s.total_rays;
    acc.inner_visit += s.inner_visit;
    acc.inner_rays += s.inner_rays;
    acc.leaf_visit += s.leaf_visit;
    acc.leaf_rays += s.leaf_rays;
    acc.leaf_objs += s.leaf_objs;
    acc.obj_intersect += s.obj_intersect;
}

/// The defaults of src/args.rs:24-37 for a frame of the size src/main.rs:103-104 hard-codes.
pub fn default_params() -> RenderParams {
    RenderParams {
        width: 1920,
        height: 1080,
        samples: 128,
        sample_offset: 0,
        max_bounces: 2,
        tile_size: 16,
        shard_index: 0,
        shard_count: 1,
        samples_per_pass: 0,
        flags: 0,
        seed: 1235468,
    }
}
