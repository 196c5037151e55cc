"""The one output of the reference itself that exists -- the example render in its repository (image.png, README.md:4)
-- against the oracle and against the GPU path.  The image's scene was random, but two regions do not depend on the
random spheres (tests/golden/make_reference_image_fixture.py says why), and a third one follows from the three fixed
spheres: the sky above the horizon must come out the same byte for byte, up to one step where a sample mean sits on a
byte boundary; so must the part of the large metal sphere (fuzz 0) that mirrors nothing but sky; the mean colour of the
ground far outside the field of small spheres must agree within one step of 256.  Together they pin, against the
reference's own numbers, Config::camera, get_ray, the sky branch and the recursion of ray_colour, Sphere::hit (on the
ground sphere and on a unit sphere: hit point and normal), HitRecord::new, the Metal scatter with vector::reflect, the
Lambertian scatter with its unit-sphere sampler, and write_colour.

A fourth region pins the lens: the horizon, 63 units away where the camera is focused at 10, is smeared over a dozen rows by
the defocus disk (src/camera.rs:68-70,127-133, vector::random_within_unit_disk src/vector.rs:29-40).  Where ground far from
the field meets the sky, the profile of the row means across that edge is the edge-spread function of the lens sampler; it
must match the reference's image to within a level of 256, where a pinhole camera is 28 levels off."""
import os

import numpy as np
import pytest
from scipy.ndimage import binary_erosion

import raytracer_amd as rt

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS, WIDTH = 150, 830


@pytest.fixture(scope="module")
def reference_rows():
    f = np.load(os.path.join(HERE, "golden", "reference_image_rows.npz"))
    assert int(f["width"]) == WIDTH and int(f["height"]) == 466 and f["rgb"].shape == (ROWS, WIDTH, 3)
    return f["rgb"].astype(np.int64)


def regions(camera, trace):
    """Masks of the two regions from the closest hits of the rays through the pixel centres."""
    jj, ii = np.meshgrid(np.arange(ROWS), np.arange(WIDTH), indexing="ij")
    d = camera["pixel00_loc"] + ii[..., None] * camera["pixel_delta_u"] + jj[..., None] * camera["pixel_delta_v"] - camera["center"]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(camera["center"], d.shape).reshape(-1, 3)
    hit = trace(o, d.reshape(-1, 3))
    prim, t = hit["prim"].reshape(ROWS, WIDTH), hit["t"].reshape(ROWS, WIDTH)
    with np.errstate(invalid="ignore"):
        p = camera["center"] + np.where(np.isfinite(t), t, 0.0)[..., None] * d
    # clear of the horizon by more than the lens can tilt a ray (the defocus disk is 0.05 wide at 10 units)
    sky = (prim < 0) & (d[..., 1] > 0.02)
    # the small spheres lie within |x|, |z| < 12 and below y = 0.4; sphere 0 is the ground
    far_ground = (prim == 0) & ((np.abs(p[..., 0]) > 13.5) | (np.abs(p[..., 2]) > 13.5))
    # sphere 3 is the large metal sphere (src/main.rs:79-82: albedo (0.7, 0.6, 0.5), fuzz 0): where its mirror image of the
    # pixel centre's ray leaves upwards and meets nothing (the small spheres end at y = 0.4, below every such ray)
    n = hit["normal"].reshape(ROWS, WIDTH, 3)
    mirrored = d - 2.0 * (d * n).sum(axis=-1, keepdims=True) * n
    on_metal = prim == 3
    second = trace(p[on_metal], mirrored[on_metal])
    metal_sky = np.zeros((ROWS, WIDTH), dtype=bool)
    metal_sky[on_metal] = (second["prim"] < 0) & (mirrored[on_metal][:, 1] > 0.05)
    metal_sky = binary_erosion(metal_sky, iterations=3)  # whole pixel footprints (jitter, lens) inside the region
    return sky, far_ground, metal_sky


def check(rendered_bytes, reference_rows, sky, far_ground, metal_sky):
    got = np.asarray(rendered_bytes).reshape(-1, WIDTH, 3)[:ROWS].astype(np.int64)
    diff = got - reference_rows
    assert sky.sum() > 5000 and far_ground.sum() > 8000 and metal_sky.sum() > 15000
    for region in (sky, metal_sky):
        assert np.abs(diff[region]).max() <= 1
        assert (diff[region] == 0).all(axis=-1).mean() >= 0.9
    mean_diff = diff[far_ground].mean(axis=0)
    assert np.abs(mean_diff).max() <= 1.0, mean_diff
    return (diff[sky] == 0).all(axis=-1).mean(), (diff[metal_sky] == 0).all(axis=-1).mean(), mean_diff


H0, H1 = 90, 121  # rows around the horizon (it lies on rows 104-106); further down the field of small spheres comes into view


def horizon_columns(camera, trace):
    """Columns whose rows H0..H1 show only sky and ground far from the field, a dozen pixels clear of any other column."""
    jj, ii = np.meshgrid(np.arange(H0, H1), np.arange(WIDTH), indexing="ij")
    d = camera["pixel00_loc"] + ii[..., None] * camera["pixel_delta_u"] + jj[..., None] * camera["pixel_delta_v"] - camera["center"]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(camera["center"], d.shape).reshape(-1, 3)
    hit = trace(o, d.reshape(-1, 3))
    prim, t = hit["prim"].reshape(d.shape[:2]), hit["t"].reshape(d.shape[:2])
    with np.errstate(invalid="ignore"):
        p = camera["center"] + np.where(np.isfinite(t), t, 0.0)[..., None] * d
    clean = (prim < 0) | ((prim == 0) & ((np.abs(p[..., 0]) > 13.5) | (np.abs(p[..., 2]) > 13.5)))
    return binary_erosion(clean.all(axis=0), iterations=12)


def check_horizon(with_lens, pinhole, reference_rows, cols):
    assert cols.sum() > 200
    ref = reference_rows[H0:H1][:, cols].mean(axis=1)
    a = np.asarray(with_lens).reshape(-1, WIDTH, 3)[H0:H1][:, cols].astype(float).mean(axis=1)
    b = np.asarray(pinhole).reshape(-1, WIDTH, 3)[H0:H1][:, cols].astype(float).mean(axis=1)
    edge = slice(96 - H0, 114 - H0)  # the smeared edge itself: rows 96 .. 113
    worst, mean = np.abs(a - ref)[edge].max(), np.abs(a - ref)[edge].mean()
    assert ref[0].min() > 220 and ref[-1].max() < 185  # sky above, ground below
    assert worst <= 1.0 and mean <= 0.4, (worst, mean)
    assert np.abs(b - ref)[edge].max() >= 15.0  # the comparison can tell a lens from a pinhole
    return worst, mean, np.abs(b - ref)[edge].max()


def camera_fields(cam):
    return {k: getattr(cam, k) for k in ("center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v")}


def test_oracle_reproduces_the_reference_image(oracle, reference_rows):
    spheres, materials = oracle.scene_rtiow(1)
    scene = oracle.Scene(spheres, materials)
    cfg = rt.config_rtiow(WIDTH, 32, 50).record().view(oracle.CONFIG_DTYPE)
    _, cam = oracle.camera_derive(cfg)
    sky, far_ground, metal_sky = regions(cam, lambda o, d: scene.trace_batch(o, d, want_p=False))
    rgb, _ = scene.render(cfg, 2024, 0, 32, tile=(0, 0, WIDTH, ROWS), nthreads=os.cpu_count() or 1)
    exact, exact_metal, mean_diff = check(oracle.quantise(rgb, 32), reference_rows, sky, far_ground, metal_sky)
    print("oracle vs image.png: %.1f %% of %d sky pixels and %.1f %% of %d mirrored-sky pixels identical, far ground mean differs by %s of 256"
          % (100 * exact, sky.sum(), 100 * exact_metal, metal_sky.sum(), np.round(mean_diff, 2)))


def test_oracle_reproduces_the_horizon_of_the_reference_image(oracle, reference_rows):
    spheres, materials = oracle.scene_rtiow(1)
    scene = oracle.Scene(spheres, materials)
    out = []
    for defocus in (0.6, 0.0):  # src/main.rs:94, and a pinhole for contrast
        cfg = rt.config_rtiow(WIDTH, 64, 50).defocus_angle(defocus).record().view(oracle.CONFIG_DTYPE)
        rgb, _ = scene.render(cfg, 5, 0, 64, tile=(0, H0, WIDTH, H1), nthreads=os.cpu_count() or 1)
        out.append(oracle.quantise(rgb, 64))
    _, cam = oracle.camera_derive(rt.config_rtiow(WIDTH, 64, 50).record().view(oracle.CONFIG_DTYPE))
    cols = horizon_columns(cam, lambda o, d: scene.trace_batch(o, d, want_p=False))
    worst, mean, pin = check_horizon(out[0], out[1], reference_rows, cols)
    print("oracle vs image.png across the horizon (%d columns): row means within %.2f of 256 (mean %.2f); a pinhole is %.1f off" % (cols.sum(), worst, mean, pin))


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_image(reference_rows):
    spheres, materials = rt.scene_rtiow(1)
    world = rt.BVH(spheres, materials)
    cam = rt.config_rtiow(WIDTH, 256, 50).camera()
    sky, far_ground, metal_sky = regions(camera_fields(cam), lambda o, d: world.trace_batch(o, d))
    rgb, _ = cam.render_sum(world, seed=2024, spp_begin=0, spp_end=256, tile=(0, 0, WIDTH, ROWS))
    exact, exact_metal, mean_diff = check(rt.quantise(rgb, 256), reference_rows, sky, far_ground, metal_sky)
    print("GPU vs image.png: %.1f %% of %d sky pixels and %.1f %% of %d mirrored-sky pixels identical, far ground mean differs by %s of 256"
          % (100 * exact, sky.sum(), 100 * exact_metal, metal_sky.sum(), np.round(mean_diff, 2)))
    # the lens: the smeared horizon
    out = []
    for defocus in (0.6, 0.0):
        c = rt.config_rtiow(WIDTH, 256, 50).defocus_angle(defocus).camera()
        rgb, _ = c.render_sum(world, seed=5, tile=(0, H0, WIDTH, H1))
        out.append(rt.quantise(rgb, 256))
    cols = horizon_columns(camera_fields(cam), lambda o, d: world.trace_batch(o, d))
    worst, mean, pin = check_horizon(out[0], out[1], reference_rows, cols)
    print("GPU vs image.png across the horizon (%d columns): row means within %.2f of 256 (mean %.2f); a pinhole is %.1f off" % (cols.sum(), worst, mean, pin))
    world.close()
