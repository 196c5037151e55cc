"""ctypes front end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by the smoke check and by the
CPU-baseline leg of bench.py; never by the shipped package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

SPHERE_DTYPE = np.dtype([("c", "<f8", 3), ("r", "<f8"), ("mat", "<u4"), ("pad", "<u4")])
MATERIAL_DTYPE = np.dtype([("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", 3), ("param", "<f8")])
CONFIG_DTYPE = np.dtype(
    [
        ("aspect_ratio", "<f8"),
        ("image_width", "<u4"),
        ("pad", "<u4"),
        ("samples_per_pixel", "<u8"),
        ("max_depth", "<u8"),
        ("vfov", "<f8"),
        ("lookfrom", "<f8", 3),
        ("lookat", "<f8", 3),
        ("vup", "<f8", 3),
        ("defocus_angle", "<f8"),
        ("focus_dist", "<f8"),
    ]
)

SAMPLER_DIRECT = 0
SAMPLER_REJECTION = 1


def build():
    """Compile liboracle.so with the Makefile next to this file."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    return C.CDLL(_LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_bvh_node_count.restype = C.c_size_t
        _lib.orc_bvh_leaf_index_count.restype = C.c_size_t
        _lib.orc_aabb_surface_area.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, 3))


def make_config(**kw):
    """Config with the reference's defaults (src/camera.rs:15-26)."""
    cfg = np.zeros(1, dtype=CONFIG_DTYPE)
    cfg["aspect_ratio"] = 16.0 / 9.0
    cfg["image_width"] = 400
    cfg["samples_per_pixel"] = 10
    cfg["max_depth"] = 10
    cfg["vfov"] = 90.0
    cfg["lookfrom"] = (0.0, 0.0, -1.0)
    cfg["lookat"] = (0.0, 0.0, 0.0)
    cfg["vup"] = (0.0, 1.0, 0.0)
    cfg["defocus_angle"] = 0.0
    cfg["focus_dist"] = 10.0
    for k, v in kw.items():
        cfg[k] = v
    return cfg


def philox4x32_10(key, ctr):
    key = np.asarray(key, dtype=np.uint32)
    ctr = np.asarray(ctr, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(key), _p(ctr), _p(out))
    return out


def philox4x32(key, ctr, rounds):
    """Philox4x32 with the given number of rounds (10: the published default; 7: what the render loop draws from)."""
    key = np.asarray(key, dtype=np.uint32)
    ctr = np.asarray(ctr, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32(_p(key), _p(ctr), _p(out), C.c_int(rounds))
    return out


def philox_pair(seed, pixel, sample, slot, domain):
    u = np.zeros(2)
    lib().orc_philox_pair(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(slot), C.c_uint32(domain), _p(u))
    return u


# ---- AABB ----
def aabb_new():
    mn, mx = np.zeros(3), np.zeros(3)
    lib().orc_aabb_new(_p(mn), _p(mx))
    return mn, mx


def aabb_is_empty(mn, mx):
    mn, mx = _f3(mn), _f3(mx)
    return bool(lib().orc_aabb_is_empty(_p(mn), _p(mx)))


def aabb_surface_area(mn, mx):
    mn, mx = _f3(mn), _f3(mx)
    return lib().orc_aabb_surface_area(_p(mn), _p(mx))


def aabb_centroid(mn, mx):
    mn, mx = _f3(mn), _f3(mx)
    c = np.zeros(3)
    lib().orc_aabb_centroid(_p(mn), _p(mx), _p(c))
    return c


def aabb_bounding_box(points):
    pts = _f3(points)
    mn, mx = np.zeros(3), np.zeros(3)
    lib().orc_aabb_bounding_box(_p(pts), C.c_size_t(len(pts)), _p(mn), _p(mx))
    return mn, mx


def _boxes(boxes):
    return np.ascontiguousarray(np.asarray([np.concatenate([np.asarray(a, float), np.asarray(b, float)]) for a, b in boxes]).reshape(-1, 6))


def aabb_union(boxes):
    b = _boxes(boxes) if len(boxes) else np.zeros((0, 6))
    mn, mx = np.zeros(3), np.zeros(3)
    lib().orc_aabb_union(_p(b), C.c_size_t(len(b)), _p(mn), _p(mx))
    return mn, mx


def aabb_intersection(boxes):
    b = _boxes(boxes) if len(boxes) else np.zeros((0, 6))
    mn, mx = np.zeros(3), np.zeros(3)
    lib().orc_aabb_intersection(_p(b), C.c_size_t(len(b)), _p(mn), _p(mx))
    return mn, mx


def slab_batch(boxes, org, direction, interval):
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float64).reshape(-1, 6))
    org, direction = _f3(org), _f3(direction)
    interval = np.ascontiguousarray(np.asarray(interval, dtype=np.float64).reshape(-1, 2))
    n = len(boxes)
    hit = np.zeros(n, dtype=np.uint8)
    lib().orc_slab_batch(_p(boxes), _p(org), _p(direction), _p(interval), C.c_size_t(n), _p(hit))
    return hit.astype(bool)


def ray_aabb_intersection(mn, mx, origin, direction, tmin, tmax):
    return bool(slab_batch([np.concatenate([mn, mx])], [origin], [direction], [[tmin, tmax]])[0])


# ---- Sphere ----
def sphere_aabb(c, r):
    c = _f3(c)
    mn, mx = np.zeros(3), np.zeros(3)
    lib().orc_sphere_aabb(_p(c), C.c_double(r), _p(mn), _p(mx))
    return mn, mx


def sphere_clipped_aabb(c, r, axis, t1, t2):
    c = _f3(c)
    mn, mx = np.zeros(3), np.zeros(3)
    lib().orc_sphere_clipped_aabb(_p(c), C.c_double(r), C.c_int(axis), C.c_double(t1), C.c_double(t2), _p(mn), _p(mx))
    return mn, mx


def sphere_hit_batch(c, r, org, direction, tmin, tmax):
    c, org, direction = _f3(c), _f3(org), _f3(direction)
    n = len(org)
    hit = np.zeros(n, dtype=np.uint8)
    t = np.full(n, np.inf)
    p = np.zeros((n, 3))
    normal = np.zeros((n, 3))
    front = np.zeros(n, dtype=np.uint8)
    lib().orc_sphere_hit_batch(_p(c), C.c_double(r), _p(org), _p(direction), C.c_size_t(n), C.c_double(tmin), C.c_double(tmax),
                               _p(hit), _p(t), _p(p), _p(normal), _p(front))
    return hit.astype(bool), t, p, normal, front.astype(bool)


# ---- Scene ----
class Scene:
    """BVH::new over flat sphere/material arrays (src/bvh.rs:44-48)."""

    def __init__(self, spheres, materials):
        self.spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self._h = C.c_void_p(lib().orc_scene_create(_p(self.spheres), C.c_size_t(len(self.spheres)), _p(self.materials),
                                                    C.c_size_t(len(self.materials))))

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh_dump(self):
        n = lib().orc_bvh_node_count(self._h)
        nl = lib().orc_bvh_leaf_index_count(self._h)
        boxes = np.zeros((n, 6))
        kind = np.zeros(n, dtype=np.int32)
        soc = np.zeros(n, dtype=np.uint64)
        leaf = np.zeros(nl, dtype=np.uint64)
        lib().orc_bvh_dump(self._h, _p(boxes), _p(kind), _p(soc), _p(leaf))
        return boxes, kind, soc, leaf

    def trace_batch(self, org, direction, tmin=1e-3, tmax=np.inf, want_p=False):
        org, direction = _f3(org), _f3(direction)
        n = len(org)
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        normal = np.zeros((n, 3))
        front = np.zeros(n, dtype=np.uint8)
        p = np.zeros((n, 3))
        stats = np.zeros(3, dtype=np.uint64)
        lib().orc_trace_batch(self._h, _p(org), _p(direction), C.c_size_t(n), C.c_double(tmin), C.c_double(tmax), _p(prim), _p(t),
                              _p(p), _p(normal), _p(front), _p(stats))
        out = dict(prim=prim, t=t, normal=normal, front=front.astype(bool), stats=stats)
        if want_p:
            out["p"] = p
        return out

    def render(self, cfg, seed, s0, s1, tile=None, mode=SAMPLER_DIRECT, nthreads=1, rgb_sum=None):
        cfg = np.ascontiguousarray(cfg, dtype=CONFIG_DTYPE)
        h = camera_derive(cfg)[0]
        w = int(cfg["image_width"][0])
        if tile is None:
            tile = (0, 0, w, h)
        tile = np.asarray(tile, dtype=np.uint32)
        if rgb_sum is None:
            rgb_sum = np.zeros((h, w, 3))
        stats = np.zeros(3, dtype=np.uint64)
        lib().orc_render(self._h, _p(cfg), C.c_uint64(seed), C.c_uint32(s0), C.c_uint32(s1), _p(tile), C.c_int(mode),
                         C.c_int(nthreads), _p(rgb_sum), _p(stats))
        return rgb_sum, stats


def scatter_batch(materials, mat_index, in_dir, p, normal, front, u):
    materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
    mat_index = np.ascontiguousarray(mat_index, dtype=np.uint32)
    in_dir, p, normal = _f3(in_dir), _f3(p), _f3(normal)
    front = np.ascontiguousarray(front, dtype=np.uint8)
    u = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1, 2))
    n = len(mat_index)
    scattered = np.zeros(n, dtype=np.uint8)
    att = np.zeros((n, 3))
    out_dir = np.zeros((n, 3))
    lib().orc_scatter_batch(_p(materials), _p(mat_index), _p(in_dir), _p(p), _p(normal), _p(front), _p(u), C.c_size_t(n),
                            _p(scattered), _p(att), _p(out_dir))
    return scattered.astype(bool), att, out_dir


def unit_from_uniforms(u):
    u = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1, 2))
    out = np.zeros((len(u), 3))
    lib().orc_unit_from_uniforms(_p(u), C.c_size_t(len(u)), _p(out))
    return out


def disk_from_uniforms(u):
    u = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1, 2))
    out = np.zeros((len(u), 3))
    lib().orc_disk_from_uniforms(_p(u), C.c_size_t(len(u)), _p(out))
    return out


def unit_rejection(seed, n):
    out = np.zeros((n, 3))
    lib().orc_unit_rejection(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


def disk_rejection(seed, n):
    out = np.zeros((n, 3))
    lib().orc_disk_rejection(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


def reflect(v, n):
    v, n = _f3(v), _f3(n)
    out = np.zeros(3)
    lib().orc_reflect(_p(v), _p(n), _p(out))
    return out


def refract(v, n, ratio):
    v, n = _f3(v), _f3(n)
    out = np.zeros(3)
    lib().orc_refract(_p(v), _p(n), C.c_double(ratio), _p(out))
    return out


def near_zero(v):
    v = _f3(v)
    return bool(lib().orc_near_zero(_p(v)))


def camera_derive(cfg):
    cfg = np.ascontiguousarray(cfg, dtype=CONFIG_DTYPE)
    h = C.c_uint32(0)
    out = np.zeros(18)
    lib().orc_camera_derive(_p(cfg), C.byref(h), _p(out))
    names = ["center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v", "defocus_disk_u", "defocus_disk_v"]
    return h.value, {k: out[3 * i:3 * i + 3].copy() for i, k in enumerate(names)}


def primary_rays(cfg, seed, ijs):
    cfg = np.ascontiguousarray(cfg, dtype=CONFIG_DTYPE)
    ijs = np.ascontiguousarray(np.asarray(ijs, dtype=np.uint32).reshape(-1, 3))
    n = len(ijs)
    org, direction = np.zeros((n, 3)), np.zeros((n, 3))
    lib().orc_primary_rays(_p(cfg), C.c_uint64(seed), _p(ijs), C.c_size_t(n), _p(org), _p(direction))
    return org, direction


def quantise(rgb_sum, spp):
    rgb = np.ascontiguousarray(np.asarray(rgb_sum, dtype=np.float64).reshape(-1, 3))
    out = np.zeros((len(rgb), 3), dtype=np.uint8)
    lib().orc_quantise(_p(rgb), C.c_size_t(len(rgb)), C.c_uint64(spp), _p(out))
    return out


def write_ppm(path, rgb_sum, w, h, spp):
    rgb = np.ascontiguousarray(np.asarray(rgb_sum, dtype=np.float64).reshape(-1, 3))
    assert len(rgb) == w * h
    rc = lib().orc_write_ppm(path.encode(), _p(rgb), C.c_uint32(w), C.c_uint32(h), C.c_uint64(spp))
    if rc:
        raise OSError("cannot write " + path)


# ---- Scenes: the oracle's own restatement of src/main.rs:42-82 and of the stress fields ----
# Written in numpy on purpose: it shares no code with the library's generator (csrc/rt_host.cpp), so that the
# fixtures and parity tests do not take the scene from the code under test.
DOM_SCENE = 3


def philox_pairs(seed, pixel, sample, slot, domain):
    """Philox4x32-10 (Salmon et al. 2011) over arrays: key = seed, counter = (pixel, sample, slot, domain).
    Returns two arrays of uniforms in [0, 1): (u64 >> 11) * 2^-53 of output words (0, 1) and (2, 3)."""
    pixel = np.asarray(pixel, dtype=np.uint64)
    c = [pixel & np.uint64(0xFFFFFFFF), np.full_like(pixel, sample), np.full_like(pixel, slot), np.full_like(pixel, domain)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    scale = 1.0 / 9007199254740992.0
    u0 = (((c[1] << np.uint64(32)) | c[0]) >> np.uint64(11)).astype(np.float64) * scale
    u1 = (((c[3] << np.uint64(32)) | c[2]) >> np.uint64(11)).astype(np.float64) * scale
    return u0, u1


def _object_draws(seed, ids, n_draws):
    """Uniform number k of object `id` is half k % 2 of the Philox block (id, 0, k // 2, DOM_SCENE)."""
    d = []
    for slot in range((n_draws + 1) // 2):
        u0, u1 = philox_pairs(seed, ids, 0, slot, DOM_SCENE)
        d += [u0, u1]
    return d


def _random_materials(seed, ids, materials, p_lambertian, p_metal, max_fuzz):
    """The three material kinds of src/main.rs:64-79 for objects `ids`: draw 2 picks the kind
    (`gen_range(0..3)` is p = 1/3 each); Lambertian albedo = random::<DVec3>() * random::<DVec3>() (draws 4..6 times
    draws 7..9, src/main.rs:67); Metal albedo in [0.5, 1]^3 and fuzz in [0, max_fuzz) (src/main.rs:72-73);
    otherwise Dielectric with ir 1.5 (src/main.rs:77)."""
    d = _object_draws(seed, ids, 10)
    lam = d[2] < p_lambertian
    met = ~lam & (d[2] < p_lambertian + p_metal)
    die = ~lam & ~met
    typ = np.where(lam, 0, np.where(met, 1, 2)).astype(np.uint32)
    albedo = np.zeros((len(ids), 3))
    for k in range(3):
        albedo[:, k] = np.where(lam, d[4 + k] * d[7 + k], np.where(met, 0.5 + 0.5 * d[4 + k], 0.0))
    materials["type"][ids] = typ
    materials["albedo"][ids] = albedo
    materials["param"][ids] = np.where(met, max_fuzz * d[7], np.where(die, 1.5, 0.0))
    return d


def scene_rtiow(seed=1):
    """src/main.rs:42-82: ground, three big spheres, then the 22 x 22 grid of small ones, every one pushed
    (src/main.rs:80); object n uses material n."""
    spheres = np.zeros(488, dtype=SPHERE_DTYPE)
    materials = np.zeros(488, dtype=MATERIAL_DTYPE)
    spheres["mat"] = np.arange(488)
    fixed = [((0.0, -1000.0, 0.0), 1000.0, 0, (0.5, 0.5, 0.5), 0.0),   # src/main.rs:45-47
             ((0.0, 1.0, 0.0), 1.0, 2, (0.0, 0.0, 0.0), 1.5),           # src/main.rs:49
             ((-4.0, 1.0, 0.0), 1.0, 0, (0.4, 0.2, 0.1), 0.0),          # src/main.rs:51-52
             ((4.0, 1.0, 0.0), 1.0, 1, (0.7, 0.6, 0.5), 0.0)]           # src/main.rs:54-55
    for k, (c, r, t, alb, param) in enumerate(fixed):
        spheres[k]["c"], spheres[k]["r"] = c, r
        materials[k]["type"], materials[k]["albedo"], materials[k]["param"] = t, alb, param
    ids = np.arange(4, 488)
    grid = ids - 4
    a, b = grid // 22 - 11, grid % 22 - 11                              # for a in -11..11 { for b in -11..11
    d = _random_materials(seed, ids, materials, 1.0 / 3.0, 1.0 / 3.0, 1.0)
    spheres["c"][ids, 0] = a + 0.9 * d[0]                               # src/main.rs:61
    spheres["c"][ids, 1] = 0.2
    spheres["c"][ids, 2] = b + 0.9 * d[1]
    spheres["r"][ids] = 0.2
    return spheres, materials


def scene_field(seed, n_small, p_lambertian, p_metal, max_fuzz=1.0):
    """The stress fields: the ground sphere of src/main.rs:46 and n_small spheres of radius 0.2 resting on it, on a
    square grid of spacing 1 with the reference's jitter of 0.9 (src/main.rs:61), materials drawn as in
    src/main.rs:64-79 with the given probabilities."""
    R = 1000.0
    n = n_small + 1
    spheres = np.zeros(n, dtype=SPHERE_DTYPE)
    materials = np.zeros(n, dtype=MATERIAL_DTYPE)
    spheres["mat"] = np.arange(n)
    spheres[0]["c"], spheres[0]["r"] = (0.0, -R, 0.0), R
    materials[0]["albedo"] = (0.5, 0.5, 0.5)
    if n_small == 0:
        return spheres, materials
    side = int(np.ceil(np.sqrt(float(n_small))))
    half = float(side // 2)
    ids = np.arange(1, n)
    k = ids - 1
    d = _random_materials(seed, ids, materials, p_lambertian, p_metal, max_fuzz)
    x = (k % side).astype(np.float64) - half + 0.9 * d[0]
    z = (k // side).astype(np.float64) - half + 0.9 * d[1]
    spheres["c"][ids, 0] = x
    spheres["c"][ids, 1] = np.sqrt(np.maximum(R * R - x * x - z * z, 0.0)) - R + 0.2
    spheres["c"][ids, 2] = z
    spheres["r"][ids] = 0.2
    return spheres, materials
