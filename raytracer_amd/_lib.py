"""ctypes binding of libraytracer_amd.so (the C ABI declared in include/rt.h).

There is no fallback: if the shared library is missing, loading raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAYTRACER_AMD_LIB names another build of the same library (used to compare kernel variants in one run).
LIB_PATH = os.environ.get("RAYTRACER_AMD_LIB") or os.path.join(_HERE, "libraytracer_amd.so")
# The product library plus the measurement harnesses (include/rt_experiments.h); tools that need them start with
# RAYTRACER_AMD_LIB set to this path.
EXPERIMENTS_LIB_PATH = os.path.join(_HERE, "libraytracer_amd_experiments.so")

SPHERE_DTYPE = np.dtype([("center", "<f8", 3), ("radius", "<f8"), ("material", "<u4"), ("reserved", "<u4")])
MATERIAL_DTYPE = np.dtype([("type", "<u4"), ("reserved", "<u4"), ("albedo", "<f8", 3), ("param", "<f8")])
CONFIG_DTYPE = np.dtype(
    [
        ("aspect_ratio", "<f8"),
        ("image_width", "<u4"),
        ("reserved", "<u4"),
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
CAMERA_DTYPE = np.dtype(
    [
        ("image_height", "<u4"),
        ("reserved", "<u4"),
        ("center", "<f8", 3),
        ("pixel00_loc", "<f8", 3),
        ("pixel_delta_u", "<f8", 3),
        ("pixel_delta_v", "<f8", 3),
        ("defocus_disk_u", "<f8", 3),
        ("defocus_disk_v", "<f8", 3),
    ]
)
RECT_DTYPE = np.dtype([("x0", "<u4"), ("y0", "<u4"), ("x1", "<u4"), ("y1", "<u4")])
RENDER_STATS_DTYPE = np.dtype([("rays", "<u8"), ("samples", "<u8"), ("kernel_ms", "<f4"), ("launches", "<u4"), ("node_steps", "<u8"),
                               ("sphere_tests", "<u8")])
SCENE_INFO_DTYPE = np.dtype(
    [
        ("n_spheres", "<u8"),
        ("n_nodes", "<u8"),
        ("n_device_nodes", "<u8"),
        ("depth", "<u4"),
        ("n_nodes_in_lds", "<u4"),
        ("spheres_in_lds", "<u4"),
        ("device", "<u4"),
        ("device_bytes", "<u8"),
        ("build_ms", "<f8"),
        ("lds_bytes", "<u8"),
        ("builder", "<u4"),
        ("n_devices", "<u4"),
        ("peer_pairs", "<u4"),
        ("peer_pairs_enabled", "<u4"),
        ("peer_matrix", "<u8"),
        ("cut_entries", "<u4"),
        ("reserved", "<u4"),
    ]
)
SHARD_DTYPE = np.dtype([("x0", "<u4"), ("y0", "<u4"), ("x1", "<u4"), ("y1", "<u4"), ("row_first", "<u4"), ("row_stride", "<u4"),
                        ("spp_begin", "<u4"), ("spp_end", "<u4")])
PARTITION_TILES, PARTITION_SAMPLES = 0, 1

LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
BUILD_HOST_SAH, BUILD_DEVICE_LBVH, BUILD_AUTO, BUILD_DEVICE_SAH = 0, 1, 2, 3

# Every symbol include/rt.h declares.
EXPORTS = [
    "rt_last_error",
    "rt_device_count",
    "rt_device_set",
    "rt_device_get",
    "rt_config_default",
    "rt_camera_derive",
    "rt_scene_create",
    "rt_scene_create_ex",
    "rt_scene_create_multi",
    "rt_scene_destroy",
    "rt_scene_get_info",
    "rt_bvh_build",
    "rt_bvh_build_device",
    "rt_bvh_destroy",
    "rt_bvh_counts",
    "rt_bvh_dump",
    "rt_scene_bvh_dump",
    "rt_trace_batch",
    "rt_scatter_batch",
    "rt_slab_batch",
    "rt_primary_rays",
    "rt_render",
    "rt_render_device",
    "rt_render_shard_device",
    "rt_render_device_noalloc",
    "rt_partition",
    "rt_render_multi",
    "rt_debug_set",
    "rt_debug_take_out_first_leaf",
    "rt_quantise",
    "rt_write_ppm",
    "rt_quantise_device",
    "rt_write_ppm_bytes",
    "rt_render_ppm",
    "rt_scene_rtiow",
    "rt_scene_field",
]


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt error %d: %s" % (code, message))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                "%s is missing: build it with `make -C raytracer_amd/csrc` (or __graft_entry__.build()); "
                "there is no CPU fallback" % LIB_PATH
            )
        L = C.CDLL(LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_config_default.restype = None
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode())


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f3(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, 3))
