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
must match the reference's image to within a level of 256, where a pinhole camera is 28 levels off.

A fifth region pins the glass: the part of the large glass sphere (src/main.rs:49, index of refraction 1.5) where the pixel
centre's ray refracts into the sphere, out of it on the far side and on into the sky.  There a pixel is the sky's colour in
the direction the ray leaves in (plus a few per cent of reflections, whose share of the region's mean does not depend on
the random spheres to within a fifth of a level: two scenes of ours agree that well).  The region's mean must match the
reference's image within half a level of 256 per channel; an index of 1.4 or 1.6 in place of 1.5 moves the red mean by two
levels, in opposite directions.  That is Dielectric::scatter with vector::refract and the front-face logic of HitRecord::new
(src/material.rs:43-65, src/vector.rs:60-65, src/hit.rs:18-27) against the reference's own numbers."""
import os

import numpy as np
import pytest
from scipy.ndimage import binary_erosion

import raytracer_amd as rt

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS, WIDTH = 150, 830
ALL_ROWS = 224  # rows the fixture holds; the glass sphere's lower half reaches down to row 215


@pytest.fixture(scope="module")
def reference_rows():
    f = np.load(os.path.join(HERE, "golden", "reference_image_rows.npz"))
    assert int(f["width"]) == WIDTH and int(f["height"]) == 466 and f["rgb"].shape == (ALL_ROWS, WIDTH, 3)
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
    diff = got - reference_rows[:ROWS]
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


GLASS = 1  # the large glass sphere's index in scene_rtiow


def glass_sky(camera, trace, scatter):
    """Pixels whose centre ray meets the glass sphere at less than 60 degrees from its normal (where Schlick's reflectance is
    below 7 %), refracts in, meets nothing but the sphere's far side, refracts out and leaves into the sky, 0.03 or more above
    the horizon; three pixels clear of any other.  scatter(prim, in_dir, p, normal, front, u) -> (scattered, attenuation, out_dir)."""
    jj, ii = np.meshgrid(np.arange(ALL_ROWS), np.arange(WIDTH), indexing="ij")
    d = camera["pixel00_loc"] + ii[..., None] * camera["pixel_delta_u"] + jj[..., None] * camera["pixel_delta_v"] - camera["center"]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = d.reshape(-1, 3)
    o = np.broadcast_to(camera["center"], d.shape)
    h = trace(o, d)
    k = np.where((h["prim"] == GLASS) & (-(d * h["normal"]).sum(axis=-1) >= 0.5))[0]
    u = np.tile([0.999999, 0.5], (len(k), 1))  # a draw the reflectance never exceeds: the ray refracts (src/material.rs:58)
    p = o[k] + h["t"][k, None] * d[k]
    ok, _, inside = scatter(h["prim"][k], d[k], p, h["normal"][k], h["front"][k], u)
    h2 = trace(p, inside)
    p2 = p + np.where(np.isfinite(h2["t"]), h2["t"], 0.0)[:, None] * inside
    ok2, _, out = scatter(np.full(len(k), GLASS), inside, p2, h2["normal"], h2["front"], u)
    h3 = trace(p2, out)
    through = ok & ok2 & ((inside * h["normal"][k]).sum(axis=-1) < 0) & (h2["prim"] == GLASS) & ~h2["front"] & ((out * h2["normal"]).sum(axis=-1) < 0)
    good = through & (h3["prim"] < 0) & (out[:, 1] > 0.03)
    m = np.zeros(ALL_ROWS * WIDTH, dtype=bool)
    m[k[good]] = True
    return binary_erosion(m.reshape(ALL_ROWS, WIDTH), iterations=3)


def check_glass(render_region, reference_rows, region):
    """render_region(ir, tile) -> bytes of the frame (rows from 0) rendered with the glass sphere's index set to ir."""
    assert region.sum() > 2500
    ys, xs = np.where(region)
    tile = (int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1)
    mean_diff = {}
    for ir in (1.5, 1.4, 1.6):
        got = np.asarray(render_region(ir, tile)).reshape(-1, WIDTH, 3)[:ALL_ROWS].astype(np.int64)
        mean_diff[ir] = (got[region] - reference_rows[region]).mean(axis=0)
    assert np.abs(mean_diff[1.5]).max() <= 0.5, mean_diff
    assert mean_diff[1.4][0] >= 1.5 and mean_diff[1.6][0] <= -1.2, mean_diff  # the comparison can tell 1.5 from its neighbours
    return mean_diff


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


def test_oracle_reproduces_the_sky_seen_through_the_glass_sphere(oracle, reference_rows):
    spheres, materials = oracle.scene_rtiow(1)
    scene = oracle.Scene(spheres, materials)
    _, cam = oracle.camera_derive(rt.config_rtiow(WIDTH, 128, 50).record().view(oracle.CONFIG_DTYPE))
    region = glass_sky(cam, lambda o, d: scene.trace_batch(o, d),
                       lambda prim, in_dir, p, n, front, u: oracle.scatter_batch(materials, spheres["mat"][prim], in_dir, p, n, front, u))

    def render_region(ir, tile):
        sp, mt = oracle.scene_rtiow(1)
        mt[GLASS]["param"] = ir
        cfg = rt.config_rtiow(WIDTH, 128, 50).record().view(oracle.CONFIG_DTYPE)
        rgb, _ = oracle.Scene(sp, mt).render(cfg, 5, 0, 128, tile=tile, nthreads=os.cpu_count() or 1)
        return oracle.quantise(rgb, 128)

    md = check_glass(render_region, reference_rows, region)
    print("oracle vs image.png through the glass sphere (%d pixels): mean difference %s of 256; with index 1.4 %s, with 1.6 %s"
          % (region.sum(), np.round(md[1.5], 2), np.round(md[1.4], 2), np.round(md[1.6], 2)))


@pytest.mark.gpu
def test_gpu_reproduces_the_sky_seen_through_the_glass_sphere(reference_rows):
    spheres, materials = rt.scene_rtiow(1)
    world = rt.BVH(spheres, materials)
    cam = rt.config_rtiow(WIDTH, 512, 50).camera()
    region = glass_sky(camera_fields(cam), lambda o, d: world.trace_batch(o, d),
                       lambda prim, in_dir, p, n, front, u: world.scatter_batch(spheres["material"][prim], in_dir, p, n, front, u))
    world.close()

    def render_region(ir, tile):
        sp, mt = rt.scene_rtiow(1)
        mt[GLASS]["param"] = ir
        w = rt.BVH(sp, mt)
        rgb, _ = cam.render_sum(w, seed=5, tile=tile)
        w.close()
        return rt.quantise(rgb, 512)

    md = check_glass(render_region, reference_rows, region)
    print("GPU vs image.png through the glass sphere (%d pixels): mean difference %s of 256; with index 1.4 %s, with 1.6 %s"
          % (region.sum(), np.round(md[1.5], 2), np.round(md[1.4], 2), np.round(md[1.6], 2)))


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
