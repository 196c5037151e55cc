"""Host-side mirror of the reference's types for the render hot path.

Names and meanings follow rfsmi/raytracer: `Config` (src/camera.rs:15-26) with
one by-value setter per field (src/util.rs:1-19), `Camera` with `render`
(src/camera.rs:138), `Sphere` (src/sphere.rs:19), `Lambertian` / `Metal` /
`Dielectric` (src/material.rs:10-19), `BVH` with `hit` (src/bvh.rs:44,50),
`Ray` and `Interval` (src/ray.rs).  All computation happens in
libraytracer_amd.so on the GPU; these classes only marshal arguments.
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import (CAMERA_DTYPE, CONFIG_DTYPE, DIELECTRIC, LAMBERTIAN, MATERIAL_DTYPE, METAL, PARTITION_SAMPLES, PARTITION_TILES, RECT_DTYPE,
                   RENDER_STATS_DTYPE, SCENE_INFO_DTYPE, SHARD_DTYPE, SPHERE_DTYPE, RtError, check, f3, ptr)

_FIELDS = ["aspect_ratio", "image_width", "samples_per_pixel", "max_depth", "vfov", "lookfrom", "lookat", "vup", "defocus_angle",
           "focus_dist"]


class Config:
    """`Config::new()` and its builder setters; `get(name)` reads a field."""

    def __init__(self, _rec=None):
        if _rec is None:
            _rec = np.zeros(1, dtype=CONFIG_DTYPE)
            _lib.lib().rt_config_default(ptr(_rec))
        self._rec = _rec

    def get(self, name):
        v = self._rec[name][0]
        return v.copy() if isinstance(v, np.ndarray) else v.item()

    def _with(self, name, value):
        rec = self._rec.copy()
        rec[name] = value
        return Config(rec)

    def record(self):
        return self._rec

    def camera(self):
        """`Config::camera()` (src/camera.rs:40-82)."""
        return Camera(self)


def _setter(name):
    def set_field(self, value):
        return self._with(name, value)

    set_field.__name__ = name
    set_field.__doc__ = "By-value setter for `%s` (src/util.rs:12-16)." % name
    return set_field


for _f in _FIELDS:
    setattr(Config, _f, _setter(_f))


class Interval:
    """src/ray.rs:9-13,30-41"""

    def __init__(self, min, max):
        self.min, self.max = float(min), float(max)

    def surrounds(self, x):
        return self.min < x < self.max

    def clamp(self, x):
        return min(max(x, self.min), self.max)


class Ray:
    """src/ray.rs:3-27.  `Ray.new` keeps the direction as given; the library
    normalises it exactly as `Ray::new` does when the ray is used."""

    def __init__(self, origin, direction):
        self.origin = np.asarray(origin, dtype=np.float64)
        self.direction = np.asarray(direction, dtype=np.float64)

    new = classmethod(lambda cls, origin, direction: cls(origin, direction))


class Material:
    type = None

    def record(self):
        rec = np.zeros(1, dtype=MATERIAL_DTYPE)
        rec["type"] = self.type
        rec["albedo"] = getattr(self, "_albedo", (0.0, 0.0, 0.0))
        rec["param"] = getattr(self, "_param", 0.0)
        return rec


class Lambertian(Material):
    """src/material.rs:10-12"""
    type = LAMBERTIAN

    def __init__(self, albedo=(0.0, 0.0, 0.0)):
        self._albedo = tuple(float(x) for x in albedo)

    def albedo(self, albedo):
        return Lambertian(albedo)


class Metal(Material):
    """src/material.rs:14-17"""
    type = METAL

    def __init__(self, albedo=(0.0, 0.0, 0.0), fuzz=0.0):
        self._albedo = tuple(float(x) for x in albedo)
        self._param = float(fuzz)

    def albedo(self, albedo):
        return Metal(albedo, self._param)

    def fuzz(self, fuzz):
        return Metal(self._albedo, fuzz)


class Dielectric(Material):
    """src/material.rs:19"""
    type = DIELECTRIC

    def __init__(self, ir=1.5):
        self._param = float(ir)

    def ir(self, ir):
        return Dielectric(ir)


class Sphere:
    """`Sphere::new(center, radius, material)` (src/sphere.rs:19-25)."""

    def __init__(self, center, radius, material):
        self.center = tuple(float(x) for x in center)
        self.radius = float(radius)
        self.material = material


class HitRecord:
    """src/hit.rs:9-15; `prim` is the index of the sphere in the list given to BVH."""

    def __init__(self, front_face, normal, p, t, prim):
        self.front_face, self.normal, self.p, self.t, self.prim = front_face, normal, p, t, prim


def flatten(objects):
    """List of Sphere -> (sphere records, material records), one material per distinct object."""
    spheres = np.zeros(len(objects), dtype=SPHERE_DTYPE)
    mats, index = [], {}
    for k, s in enumerate(objects):
        key = id(s.material)
        if key not in index:
            index[key] = len(mats)
            mats.append(s.material.record())
        spheres[k]["center"] = s.center
        spheres[k]["radius"] = s.radius
        spheres[k]["material"] = index[key]
    materials = np.concatenate(mats) if mats else np.zeros(0, dtype=MATERIAL_DTYPE)
    return spheres, materials


class BVH:
    """`BVH::new(objects)` (src/bvh.rs:44-48) on the current HIP device.

    `objects` is a list of Sphere, or a pair (sphere records, material records).  builder: BUILD_HOST_SAH (the
    reference's tree, built on the host), BUILD_DEVICE_SAH (the same tree built on the GPU), BUILD_DEVICE_LBVH (built on the GPU by clustering; same hits,
    a different tree) or BUILD_AUTO (the host below 1 024 spheres, BUILD_DEVICE_SAH above: what rt_scene_create does).  devices: None for one
    copy on the current device; "all" or a list of HIP ordinals for a copy on each (Camera.render and
    Camera.render_multi then use all of them)."""

    def __init__(self, objects, materials=None, builder=0, devices=None):
        if materials is None and not isinstance(objects, np.ndarray):
            spheres, materials = flatten(list(objects))
        else:
            spheres = np.ascontiguousarray(objects, dtype=SPHERE_DTYPE)
            materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.spheres, self.materials = spheres, materials
        h = C.c_void_p()
        if devices is None:
            check(_lib.lib().rt_scene_create_ex(ptr(spheres), C.c_size_t(len(spheres)), ptr(materials), C.c_size_t(len(materials)),
                                                C.c_uint32(builder), C.byref(h)))
        else:
            devs = None if isinstance(devices, str) else np.ascontiguousarray(devices, dtype=np.int32)
            check(_lib.lib().rt_scene_create_multi(ptr(spheres), C.c_size_t(len(spheres)), ptr(materials), C.c_size_t(len(materials)),
                                                   C.c_uint32(builder), ptr(devs), C.c_size_t(0 if devs is None else len(devs)), C.byref(h)))
        self._h = h

    new = classmethod(lambda cls, objects: cls(objects))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        rec = np.zeros(1, dtype=SCENE_INFO_DTYPE)
        check(_lib.lib().rt_scene_get_info(self._h, ptr(rec)))
        return {k: rec[k][0].item() for k in SCENE_INFO_DTYPE.names}

    def dump(self):
        n = self.info()["n_nodes"]
        boxes = np.zeros((n, 6))
        kind = np.zeros(n, dtype=np.int32)
        soc = np.zeros(n, dtype=np.uint64)
        leaf = np.zeros(len(self.spheres), dtype=np.uint64)
        check(_lib.lib().rt_scene_bvh_dump(self._h, ptr(boxes), ptr(kind), ptr(soc), ptr(leaf)))
        return boxes, kind, soc, leaf

    def trace_batch(self, org, direction, t_min=1e-3, t_max=np.inf):
        """`BVH::hit` for n rays at once."""
        org, direction = f3(org), f3(direction)
        n = len(org)
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        normal = np.zeros((n, 3))
        front = np.zeros(n, dtype=np.uint8)
        check(_lib.lib().rt_trace_batch(self._h, ptr(org), ptr(direction), C.c_size_t(n), C.c_double(t_min), C.c_double(t_max),
                                        ptr(prim), ptr(t), ptr(normal), ptr(front)))
        return dict(prim=prim, t=t, normal=normal, front=front.astype(bool))

    def hit(self, r, ray_t):
        """`BVH::hit(&Ray, Interval) -> Option<HitRecord>` (src/bvh.rs:50)."""
        out = self.trace_batch([r.origin], [r.direction], ray_t.min, ray_t.max)
        if out["prim"][0] < 0:
            return None
        d = r.direction / np.sqrt((r.direction ** 2).sum())
        return HitRecord(bool(out["front"][0]), out["normal"][0], r.origin + out["t"][0] * d, float(out["t"][0]), int(out["prim"][0]))

    def scatter_batch(self, material, in_dir, p, normal, front, u):
        """`Material::scatter` for n events with explicit uniforms u (n x 2)."""
        material = np.ascontiguousarray(material, dtype=np.uint32)
        in_dir, p, normal = f3(in_dir), f3(p), f3(normal)
        front = np.ascontiguousarray(front, dtype=np.uint8)
        u = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1, 2))
        n = len(material)
        scattered = np.zeros(n, dtype=np.uint8)
        att = np.zeros((n, 3))
        out_dir = np.zeros((n, 3))
        check(_lib.lib().rt_scatter_batch(self._h, ptr(material), ptr(in_dir), ptr(p), ptr(normal), ptr(front), ptr(u), C.c_size_t(n),
                                          ptr(scattered), ptr(att), ptr(out_dir)))
        return scattered.astype(bool), att, out_dir


class Camera:
    """Derived camera (src/camera.rs:28-37) with `render` (src/camera.rs:138-165)."""

    def __init__(self, config):
        self.config = config
        rec = np.zeros(1, dtype=CAMERA_DTYPE)
        check(_lib.lib().rt_camera_derive(ptr(config.record()), ptr(rec)))
        self._rec = rec
        self.image_height = int(rec["image_height"][0])
        for k in ("center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v", "defocus_disk_u", "defocus_disk_v"):
            setattr(self, k, rec[k][0].copy())

    def primary_rays(self, ijs, seed=0):
        """`Camera::get_ray(i, j)` for (i, j, sample) triples."""
        ijs = np.ascontiguousarray(np.asarray(ijs, dtype=np.uint32).reshape(-1, 3))
        n = len(ijs)
        org, direction = np.zeros((n, 3)), np.zeros((n, 3))
        check(_lib.lib().rt_primary_rays(ptr(self.config.record()), C.c_uint64(seed), ptr(ijs), C.c_size_t(n), ptr(org), ptr(direction)))
        return org, direction

    def render_sum(self, bvh, seed=0, spp_begin=0, spp_end=None, tile=None, rgb_sum=None):
        """The parallel part of `render`: per-pixel radiance sums (H x W x 3) and the call's stats."""
        w, h = self.config.get("image_width"), self.image_height
        if spp_end is None:
            spp_end = self.config.get("samples_per_pixel")
        if rgb_sum is None:
            rgb_sum = np.zeros((h, w, 3))
        rect = None
        if tile is not None:
            rect = np.zeros(1, dtype=RECT_DTYPE)
            rect["x0"], rect["y0"], rect["x1"], rect["y1"] = tile
        stats = np.zeros(1, dtype=RENDER_STATS_DTYPE)
        check(_lib.lib().rt_render(bvh._h, ptr(self.config.record()), C.c_uint64(seed), C.c_uint32(spp_begin), C.c_uint32(spp_end),
                                   ptr(rect), ptr(rgb_sum), ptr(stats)))
        return rgb_sum, {k: stats[k][0].item() for k in RENDER_STATS_DTYPE.names}

    def render_device(self, bvh, rgb_sum_ptr, seed=0, spp_begin=0, spp_end=None, tile=None, stream=None):
        """As render_sum, adding into a device buffer (an integer device address) on `stream`."""
        if spp_end is None:
            spp_end = self.config.get("samples_per_pixel")
        rect = None
        if tile is not None:
            rect = np.zeros(1, dtype=RECT_DTYPE)
            rect["x0"], rect["y0"], rect["x1"], rect["y1"] = tile
        stats = np.zeros(1, dtype=RENDER_STATS_DTYPE)
        check(_lib.lib().rt_render_device(bvh._h, ptr(self.config.record()), C.c_uint64(seed), C.c_uint32(spp_begin), C.c_uint32(spp_end),
                                          ptr(rect), C.c_void_p(rgb_sum_ptr), C.c_void_p(stream or 0), ptr(stats)))
        return {k: stats[k][0].item() for k in RENDER_STATS_DTYPE.names}

    def render_shard_device(self, bvh, rgb_sum_ptr, shard, seed=0, stream=None):
        """One share of a frame (a SHARD_DTYPE record, e.g. from `partition`) added into a device buffer."""
        stats = np.zeros(1, dtype=RENDER_STATS_DTYPE)
        shard = np.ascontiguousarray(shard, dtype=SHARD_DTYPE).reshape(1)
        check(_lib.lib().rt_render_shard_device(bvh._h, ptr(self.config.record()), C.c_uint64(seed), ptr(shard), C.c_void_p(rgb_sum_ptr),
                                                C.c_void_p(stream or 0), ptr(stats)))
        return {k: stats[k][0].item() for k in RENDER_STATS_DTYPE.names}

    def render_device_noalloc(self, bvh, rgb_sum_ptr, seed=0, spp_begin=0, spp_end=None, tile=None, stream=None, want_stats=True):
        """render_device through the entry point that allocates nothing (static device workspace)."""
        if spp_end is None:
            spp_end = self.config.get("samples_per_pixel")
        rect = None
        if tile is not None:
            rect = np.zeros(1, dtype=RECT_DTYPE)
            rect["x0"], rect["y0"], rect["x1"], rect["y1"] = tile
        stats = np.zeros(1, dtype=RENDER_STATS_DTYPE) if want_stats else None
        check(_lib.lib().rt_render_device_noalloc(bvh._h, ptr(self.config.record()), C.c_uint64(seed), C.c_uint32(spp_begin),
                                                  C.c_uint32(spp_end), ptr(rect), C.c_void_p(rgb_sum_ptr), C.c_void_p(stream or 0), ptr(stats)))
        return {k: stats[k][0].item() for k in RENDER_STATS_DTYPE.names} if want_stats else None

    def render_wavefront_device(self, bvh, rgb_sum_ptr, seed=0, spp_begin=0, spp_end=None):
        """The measurement harness: the same samples through the wavefront pipeline (whole frame).  Only
        libraytracer_amd_experiments.so has it: start the process with RAYTRACER_AMD_LIB=raytracer_amd._lib.EXPERIMENTS_LIB_PATH."""
        if not hasattr(_lib.lib(), "rt_experiment_wavefront_device"):
            raise RuntimeError("the wavefront harness is not part of %s; set RAYTRACER_AMD_LIB=%s" % (_lib.LIB_PATH, _lib.EXPERIMENTS_LIB_PATH))
        if spp_end is None:
            spp_end = self.config.get("samples_per_pixel")
        stats = np.zeros(1, dtype=RENDER_STATS_DTYPE)
        check(_lib.lib().rt_experiment_wavefront_device(bvh._h, ptr(self.config.record()), C.c_uint64(seed), C.c_uint32(spp_begin),
                                                        C.c_uint32(spp_end), C.c_void_p(rgb_sum_ptr), ptr(stats)))
        return {k: stats[k][0].item() for k in RENDER_STATS_DTYPE.names}

    def partition(self, n_devices, partition=PARTITION_TILES, spp_begin=0, spp_end=None, tile=None):
        """The shares `render_multi` gives to n_devices devices: an array of SHARD_DTYPE records."""
        if spp_end is None:
            spp_end = self.config.get("samples_per_pixel")
        rect = None
        if tile is not None:
            rect = np.zeros(1, dtype=RECT_DTYPE)
            rect["x0"], rect["y0"], rect["x1"], rect["y1"] = tile
        shards = np.zeros(n_devices, dtype=SHARD_DTYPE)
        check(_lib.lib().rt_partition(ptr(self.config.record()), C.c_uint32(spp_begin), C.c_uint32(spp_end), ptr(rect), C.c_uint32(partition),
                                      C.c_uint32(n_devices), ptr(shards)))
        return shards

    def render_multi(self, bvh, seed=0, spp_begin=0, spp_end=None, tile=None, partition=PARTITION_TILES, rgb_sum=None):
        """render_sum on every device of a scene created with devices=...: per-pixel sums on the host."""
        w, h = self.config.get("image_width"), self.image_height
        if spp_end is None:
            spp_end = self.config.get("samples_per_pixel")
        if rgb_sum is None:
            rgb_sum = np.zeros((h, w, 3))
        rect = None
        if tile is not None:
            rect = np.zeros(1, dtype=RECT_DTYPE)
            rect["x0"], rect["y0"], rect["x1"], rect["y1"] = tile
        stats = np.zeros(1, dtype=RENDER_STATS_DTYPE)
        check(_lib.lib().rt_render_multi(bvh._h, ptr(self.config.record()), C.c_uint64(seed), C.c_uint32(spp_begin), C.c_uint32(spp_end),
                                         ptr(rect), C.c_uint32(partition), ptr(rgb_sum), ptr(stats)))
        return rgb_sum, {k: stats[k][0].item() for k in RENDER_STATS_DTYPE.names}

    def render(self, bvh, seed=0, fd=1):
        """`Camera::render(&self, &BVH)`: PPM text to fd (stdout), elapsed time to stderr.  The frame stays on the
        device until it is quantised; three bytes per pixel come back."""
        stats = np.zeros(1, dtype=RENDER_STATS_DTYPE)
        check(_lib.lib().rt_render_ppm(bvh._h, ptr(self.config.record()), C.c_uint64(seed), C.c_int(fd), ptr(stats)))
        sys.stderr.write("Completed in %.3f seconds\n" % (stats["kernel_ms"][0] / 1e3))
        return {k: stats[k][0].item() for k in RENDER_STATS_DTYPE.names}


def device_count():
    n = C.c_int(0)
    check(_lib.lib().rt_device_count(C.byref(n)))
    return n.value


def set_device(ordinal):
    """Selects the HIP device of the library's own runtime (scenes are created on it).  With PyTorch in the
    process, call this with the same ordinal as torch.cuda.set_device: the wheel brings its own HIP runtime."""
    check(_lib.lib().rt_device_set(C.c_int(ordinal)))


def get_device():
    n = C.c_int(0)
    check(_lib.lib().rt_device_get(C.byref(n)))
    return n.value


def quantise(rgb_sum, samples_per_pixel):
    """`write_colour` (src/camera.rs:85-95): N x 3 bytes."""
    rgb = np.ascontiguousarray(np.asarray(rgb_sum, dtype=np.float64).reshape(-1, 3))
    out = np.zeros((len(rgb), 3), dtype=np.uint8)
    check(_lib.lib().rt_quantise(ptr(rgb), C.c_size_t(len(rgb)), C.c_uint64(samples_per_pixel), ptr(out)))
    return out


def quantise_device(rgb_sum_ptr, n_pixels, samples_per_pixel, out_ptr, stream=None):
    """`write_colour` for a frame in device memory (integer device addresses); 3 bytes per pixel into out_ptr."""
    check(_lib.lib().rt_quantise_device(C.c_void_p(rgb_sum_ptr), C.c_size_t(n_pixels), C.c_uint64(samples_per_pixel), C.c_void_p(out_ptr),
                                        C.c_void_p(stream or 0)))


def write_ppm_bytes(fd, rgb_bytes, width, height):
    rgb_bytes = np.ascontiguousarray(rgb_bytes, dtype=np.uint8)
    if rgb_bytes.size != width * height * 3:
        raise ValueError("rgb_bytes must hold width x height x 3 bytes")
    check(_lib.lib().rt_write_ppm_bytes(C.c_int(fd), ptr(rgb_bytes), C.c_uint32(width), C.c_uint32(height)))


def write_ppm(fd, rgb_sum, width, height, samples_per_pixel):
    rgb = np.ascontiguousarray(np.asarray(rgb_sum, dtype=np.float64).reshape(-1, 3))
    if len(rgb) != width * height:
        raise ValueError("rgb_sum does not have width*height pixels")
    check(_lib.lib().rt_write_ppm(C.c_int(fd), ptr(rgb), C.c_uint32(width), C.c_uint32(height), C.c_uint64(samples_per_pixel)))


def slab_batch(boxes, org, direction, interval):
    """The kernels' node test for n (box, ray, interval) triples."""
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float64).reshape(-1, 6))
    org, direction = f3(org), f3(direction)
    interval = np.ascontiguousarray(np.asarray(interval, dtype=np.float64).reshape(-1, 2))
    n = len(boxes)
    hit = np.zeros(n, dtype=np.uint8)
    check(_lib.lib().rt_slab_batch(ptr(boxes), ptr(org), ptr(direction), ptr(interval), C.c_size_t(n), ptr(hit)))
    return hit.astype(bool)


def host_bvh(spheres, on_device=False):
    """The tree alone, in the reference's layout: returns (boxes, kind, skip_or_count, leaf_indices, depth).  on_device: built by the
    GPU's builder and brought back (needs a GPU) instead of by the host's."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    h = C.c_void_p()
    L = _lib.lib()
    check((L.rt_bvh_build_device if on_device else L.rt_bvh_build)(ptr(spheres), C.c_size_t(len(spheres)), C.byref(h)))
    try:
        nn, nl, depth = C.c_uint64(), C.c_uint64(), C.c_uint32()
        check(L.rt_bvh_counts(h, C.byref(nn), C.byref(nl), C.byref(depth)))
        boxes = np.zeros((nn.value, 6))
        kind = np.zeros(nn.value, dtype=np.int32)
        soc = np.zeros(nn.value, dtype=np.uint64)
        leaf = np.zeros(nl.value, dtype=np.uint64)
        check(L.rt_bvh_dump(h, ptr(boxes), ptr(kind), ptr(soc), ptr(leaf)))
    finally:
        L.rt_bvh_destroy(h)
    return boxes, kind, soc, leaf, depth.value


def scene_rtiow(seed=1):
    """Seeded restatement of the scene of src/main.rs:42-82 as flat records."""
    spheres = np.zeros(488, dtype=SPHERE_DTYPE)
    materials = np.zeros(488, dtype=MATERIAL_DTYPE)
    check(_lib.lib().rt_scene_rtiow(C.c_uint64(seed), ptr(spheres), ptr(materials)))
    return spheres, materials


def scene_field(seed, n_small, p_lambertian, p_metal, max_fuzz=1.0):
    spheres = np.zeros(n_small + 1, dtype=SPHERE_DTYPE)
    materials = np.zeros(n_small + 1, dtype=MATERIAL_DTYPE)
    check(_lib.lib().rt_scene_field(C.c_uint64(seed), C.c_size_t(n_small), C.c_double(p_lambertian), C.c_double(p_metal),
                                    C.c_double(max_fuzz), ptr(spheres), ptr(materials)))
    return spheres, materials


def config_rtiow(image_width=830, samples_per_pixel=750, max_depth=50):
    """The camera of src/main.rs:86-96."""
    return (Config().aspect_ratio(16.0 / 9.0).image_width(image_width).vfov(20.0).lookfrom((13.0, 2.0, 3.0)).lookat((0.0, 0.0, 0.0))
            .vup((0.0, 1.0, 0.0)).defocus_angle(0.6).focus_dist(10.0).samples_per_pixel(samples_per_pixel).max_depth(max_depth))


def config_field(n_small, image_width=3840, samples_per_pixel=64, max_depth=50):
    """A camera pulled back to frame `scene_field(seed, n_small, ...)`: it looks at the middle of the field from
    above one corner, focused on the middle."""
    side = float(np.ceil(np.sqrt(n_small)))
    eye = np.array([0.55 * side, 0.2 * side + 2.0, 0.55 * side])
    return (Config().aspect_ratio(16.0 / 9.0).image_width(image_width).vfov(40.0).lookfrom(tuple(eye)).lookat((0.0, 0.0, 0.0))
            .vup((0.0, 1.0, 0.0)).defocus_angle(0.2).focus_dist(float(np.linalg.norm(eye))).samples_per_pixel(samples_per_pixel)
            .max_depth(max_depth))
