//! Safe wrapper a maintainer of the reference calls from `Camera::render`.
//!
//! The reference's primitives and materials are trait objects with private
//! fields, so the crate needs two accessors added next to the traits
//! (`trait Hit` in src/hit.rs:34-38, `trait Material` in src/material.rs:6-8):
//!
//! ```ignore
//! fn gpu_sphere(&self) -> Option<GpuSphere>;   // on Hit; Sphere returns Some
//! fn gpu_material(&self) -> GpuMaterial;       // on Material
//! ```
//!
//! With those, `Camera::render` (src/camera.rs:138-165) becomes: flatten once,
//! `Scene::new`, `Scene::render`, `write_ppm`.
pub mod ffi;

use std::ffi::CStr;
use std::ptr;

use glam::DVec3;

#[derive(Clone, Copy, Debug)]
pub struct GpuSphere {
    pub center: DVec3,
    pub radius: f64,
    /// Address of the `Arc<dyn Material>` payload: spheres sharing an Arc share a record.
    pub material_id: usize,
    pub material: GpuMaterial,
}

#[derive(Clone, Copy, Debug)]
pub enum GpuMaterial {
    Lambertian { albedo: DVec3 },
    Metal { albedo: DVec3, fuzz: f64 },
    Dielectric { ir: f64 },
}

#[derive(Debug)]
pub struct Error(pub i32, pub String);

fn check(status: i32) -> Result<(), Error> {
    if status == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::rt_last_error()) }.to_string_lossy().into_owned();
    Err(Error(status, msg))
}

impl From<GpuMaterial> for ffi::rt_material {
    fn from(m: GpuMaterial) -> Self {
        match m {
            GpuMaterial::Lambertian { albedo } => Self { kind: ffi::RT_LAMBERTIAN, albedo: albedo.to_array(), ..Default::default() },
            GpuMaterial::Metal { albedo, fuzz } => Self { kind: ffi::RT_METAL, albedo: albedo.to_array(), param: fuzz, ..Default::default() },
            GpuMaterial::Dielectric { ir } => Self { kind: ffi::RT_DIELECTRIC, param: ir, ..Default::default() },
        }
    }
}

/// The device-resident scene: what `BVH::new(objects)` owns on the GPU side.
pub struct Scene(*mut ffi::rt_scene);

// The library documents a scene as immutable after creation.
unsafe impl Send for Scene {}
unsafe impl Sync for Scene {}

impl Scene {
    /// A copy of the scene on the GPU that is current in the calling thread (`rt_device_set`): what a process that was
    /// given one GPU of a node wants.
    pub fn new(spheres: &[GpuSphere]) -> Result<Self, Error> {
        Self::create(spheres, false)
    }

    /// A copy on every GPU the process can see: one process driving the whole node (`render` then shares a frame out).
    pub fn new_on_all_devices(spheres: &[GpuSphere]) -> Result<Self, Error> {
        Self::create(spheres, true)
    }

    fn create(spheres: &[GpuSphere], all_devices: bool) -> Result<Self, Error> {
        let mut ids: Vec<usize> = Vec::new();
        let mut materials: Vec<ffi::rt_material> = Vec::new();
        let mut flat = Vec::with_capacity(spheres.len());
        for s in spheres {
            let index = match ids.iter().position(|&id| id == s.material_id) {
                Some(k) => k,
                None => {
                    ids.push(s.material_id);
                    materials.push(s.material.into());
                    ids.len() - 1
                }
            };
            flat.push(ffi::rt_sphere { center: s.center.to_array(), radius: s.radius, material: index as u32, reserved: 0 });
        }
        // A null device list means every visible GPU; else the current one alone.
        let mut current: i32 = 0;
        check(unsafe { ffi::rt_device_get(&mut current) })?;
        let list = [current];
        let mut handle = ptr::null_mut();
        check(unsafe {
            ffi::rt_scene_create_multi(
                flat.as_ptr(),
                flat.len(),
                materials.as_ptr(),
                materials.len(),
                ffi::RT_BUILD_AUTO,
                if all_devices { ptr::null() } else { list.as_ptr() },
                if all_devices { 0 } else { 1 },
                &mut handle,
            )
        })?;
        Ok(Self(handle))
    }

    /// Radiance sums, row-major `width x height x 3`, for all samples of all pixels.
    pub fn render(&self, cfg: &ffi::rt_config, image_height: u32, seed: u64) -> Result<(Vec<f64>, ffi::rt_render_stats), Error> {
        let mut sums = vec![0.0f64; cfg.image_width as usize * image_height as usize * 3];
        let mut stats = ffi::rt_render_stats::default();
        check(unsafe {
            ffi::rt_render_multi(
                self.0,
                cfg,
                seed,
                0,
                cfg.samples_per_pixel as u32,
                ptr::null(),
                ffi::RT_PARTITION_TILES,
                sums.as_mut_ptr(),
                &mut stats,
            )
        })?;
        Ok((sums, stats))
    }
}

impl Scene {
    /// All of `Camera::render` in one call: render, quantise on the device, PPM text to stdout.
    pub fn render_ppm(&self, cfg: &ffi::rt_config, seed: u64) -> Result<ffi::rt_render_stats, Error> {
        let mut stats = ffi::rt_render_stats::default();
        check(unsafe { ffi::rt_render_ppm(self.0, cfg, seed, 1, &mut stats) })?;
        Ok(stats)
    }
}

impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { ffi::rt_scene_destroy(self.0) };
    }
}

/// The `println!` loop of src/camera.rs:139,157-164 in one buffered write to stdout.
pub fn write_ppm(sums: &[f64], width: u32, height: u32, samples_per_pixel: u64) -> Result<(), Error> {
    assert_eq!(sums.len(), width as usize * height as usize * 3);
    check(unsafe { ffi::rt_write_ppm(1, sums.as_ptr(), width, height, samples_per_pixel) })
}
