"""MI355X-native render hot path of rfsmi/raytracer behind the reference's type names."""
from ._lib import (BUILD_AUTO, BUILD_DEVICE_LBVH, BUILD_DEVICE_SAH, BUILD_HOST_SAH, CAMERA_DTYPE, CONFIG_DTYPE, DIELECTRIC, EXPORTS, LAMBERTIAN, LIB_PATH, MATERIAL_DTYPE, METAL, PARTITION_SAMPLES, PARTITION_TILES, RECT_DTYPE,
                   RENDER_STATS_DTYPE, SCENE_INFO_DTYPE, SHARD_DTYPE, SPHERE_DTYPE, RtError)
from .api import (BVH, Camera, Config, Dielectric, HitRecord, Interval, Lambertian, Material, Metal, Ray, Sphere, config_field, config_rtiow, device_count, flatten, get_device,
                  host_bvh, quantise, quantise_device, scene_field, scene_rtiow, set_device, slab_batch, write_ppm, write_ppm_bytes)
