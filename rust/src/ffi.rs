//! Raw declarations of include/rt.h (the subset the render path needs).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_LAMBERTIAN: u32 = 0;
pub const RT_METAL: u32 = 1;
pub const RT_DIELECTRIC: u32 = 2;
pub const RT_BUILD_HOST_SAH: u32 = 0;
pub const RT_BUILD_DEVICE_LBVH: u32 = 1;
pub const RT_BUILD_AUTO: u32 = 2;
pub const RT_BUILD_DEVICE_SAH: u32 = 3;
pub const RT_PARTITION_TILES: u32 = 0;
pub const RT_PARTITION_SAMPLES: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_material {
    pub kind: u32,
    pub reserved: u32,
    pub albedo: [f64; 3],
    pub param: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_sphere {
    pub center: [f64; 3],
    pub radius: f64,
    pub material: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_config {
    pub aspect_ratio: f64,
    pub image_width: u32,
    pub reserved: u32,
    pub samples_per_pixel: u64,
    pub max_depth: u64,
    pub vfov: f64,
    pub lookfrom: [f64; 3],
    pub lookat: [f64; 3],
    pub vup: [f64; 3],
    pub defocus_angle: f64,
    pub focus_dist: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_rect {
    pub x0: u32,
    pub y0: u32,
    pub x1: u32,
    pub y1: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_stats {
    pub rays: u64,
    pub samples: u64,
    pub kernel_ms: f32,
    pub launches: u32,
    pub node_steps: u64,
    pub sphere_tests: u64,
}

#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count(count: *mut c_int) -> c_int;
    pub fn rt_device_set(ordinal: c_int) -> c_int;
    pub fn rt_device_get(ordinal: *mut c_int) -> c_int;
    pub fn rt_scene_create(
        spheres: *const rt_sphere,
        n_spheres: usize,
        materials: *const rt_material,
        n_materials: usize,
        out: *mut *mut rt_scene,
    ) -> c_int;
    pub fn rt_scene_create_ex(
        spheres: *const rt_sphere,
        n_spheres: usize,
        materials: *const rt_material,
        n_materials: usize,
        builder: u32,
        out: *mut *mut rt_scene,
    ) -> c_int;
    pub fn rt_scene_create_multi(
        spheres: *const rt_sphere,
        n_spheres: usize,
        materials: *const rt_material,
        n_materials: usize,
        builder: u32,
        devices: *const i32,
        n_devices: usize,
        out: *mut *mut rt_scene,
    ) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene) -> c_int;
    pub fn rt_render_multi(
        scene: *const rt_scene,
        cfg: *const rt_config,
        seed: u64,
        spp_begin: u32,
        spp_end: u32,
        tile: *const rt_rect,
        partition: u32,
        rgb_sum: *mut f64,
        stats: *mut rt_render_stats,
    ) -> c_int;
    pub fn rt_render(
        scene: *const rt_scene,
        cfg: *const rt_config,
        seed: u64,
        spp_begin: u32,
        spp_end: u32,
        tile: *const rt_rect,
        rgb_sum: *mut f64,
        stats: *mut rt_render_stats,
    ) -> c_int;
    pub fn rt_render_device(
        scene: *const rt_scene,
        cfg: *const rt_config,
        seed: u64,
        spp_begin: u32,
        spp_end: u32,
        tile: *const rt_rect,
        rgb_sum_dev: *mut f64,
        stream: *mut c_void,
        stats: *mut rt_render_stats,
    ) -> c_int;
    pub fn rt_trace_batch(
        scene: *const rt_scene,
        org: *const f64,
        dir: *const f64,
        n: usize,
        t_min: f64,
        t_max: f64,
        prim: *mut i32,
        t: *mut f64,
        normal: *mut f64,
        front_face: *mut u8,
    ) -> c_int;
    pub fn rt_render_ppm(scene: *const rt_scene, cfg: *const rt_config, seed: u64, fd: c_int, stats: *mut rt_render_stats) -> c_int;
    pub fn rt_write_ppm(fd: c_int, rgb_sum: *const f64, width: u32, height: u32, samples_per_pixel: u64) -> c_int;
}
