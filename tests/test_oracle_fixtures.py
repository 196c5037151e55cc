"""CPU-only checks of the oracle: analytic known answers for the functions the
reference has no tests for, the committed golden vectors (regression pin of the
oracle itself), and the statistical equivalence of the direct sampling maps
with the reference's rejection loops."""
import math
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _scene(oracle):
    z = np.load(os.path.join(GOLD, "scene_rtiow_seed1.npz"))
    return z["spheres"], z["materials"], oracle.Scene(z["spheres"].view(oracle.SPHERE_DTYPE), z["materials"].view(oracle.MATERIAL_DTYPE))


# ---- analytic known answers (src/sphere.rs:29-48, src/hit.rs:18-27) ----
def test_sphere_hit_through_centre(oracle):
    hit, t, p, n, front = oracle.sphere_hit_batch([0, 0, 5], 2.0, [[0, 0, 0]], [[0, 0, 3.0]], 1e-3, math.inf)
    assert hit[0] and t[0] == 3.0 and (p[0] == [0, 0, 3]).all() and (n[0] == [0, 0, -1]).all() and front[0]


def test_sphere_hit_from_inside_uses_far_root_and_flips_normal(oracle):
    hit, t, p, n, front = oracle.sphere_hit_batch([0, 0, 0], 2.0, [[0, 0, 0]], [[1.0, 0, 0]], 1e-3, math.inf)
    assert hit[0] and t[0] == 2.0 and not front[0] and (n[0] == [-1, 0, 0]).all()


def test_sphere_tangent_and_miss(oracle):
    # tangent: discriminant 0, one root at the tangent point
    hit, t, _, n, front = oracle.sphere_hit_batch([0, 1, 5], 1.0, [[0, 0, 0]], [[0, 0, 1.0]], 1e-3, math.inf)
    # the outward normal (0,-1,0) is perpendicular to the ray: `dot < 0.0` is false, so the normal is flipped (src/hit.rs:19-23)
    assert hit[0] and t[0] == 5.0 and not front[0] and np.allclose(n[0], [0, 1, 0])
    hit, *_ = oracle.sphere_hit_batch([0, 1.0000001, 5], 1.0, [[0, 0, 0]], [[0, 0, 1.0]], 1e-3, math.inf)
    assert not hit[0]


def test_sphere_interval_is_open(oracle):
    # Interval::surrounds is strict (src/ray.rs:35-37): a root exactly at t_max or t_min does not count
    hit, t, *_ = oracle.sphere_hit_batch([0, 0, 5], 2.0, [[0, 0, 0]], [[0, 0, 1.0]], 1e-3, 3.0)
    assert not hit[0]
    hit, t, *_ = oracle.sphere_hit_batch([0, 0, 5], 2.0, [[0, 0, 0]], [[0, 0, 1.0]], 3.0, 10.0)
    assert hit[0] and t[0] == 7.0


def test_negative_radius_flips_normal(oracle):
    hit, t, _, n, front = oracle.sphere_hit_batch([0, 0, 5], -2.0, [[0, 0, 0]], [[0, 0, 1.0]], 1e-3, math.inf)
    assert hit[0] and t[0] == 3.0 and not front[0]


# ---- vector.rs helpers and materials (src/vector.rs:52-65, src/material.rs:21-65) ----
def test_reflect_refract_near_zero(oracle):
    assert np.allclose(oracle.reflect([1, -1, 0], [0, 1, 0]), [1, 1, 0])
    # normal incidence passes straight through whatever the ratio
    assert np.allclose(oracle.refract([0, -1, 0], [0, 1, 0], 1.0 / 1.5), [0, -1, 0])
    # Snell: sin(out) = ratio * sin(in)
    v = np.array([math.sin(0.5), -math.cos(0.5), 0.0])
    out = oracle.refract(v, [0, 1, 0], 1.0 / 1.5)
    assert abs(out[0] - math.sin(0.5) / 1.5) < 1e-15 and abs(np.linalg.norm(out) - 1) < 1e-15
    assert oracle.near_zero([1e-6, -1e-6, 0]) and not oracle.near_zero([1.1e-6, 0, 0])


def _mat(oracle, kind, albedo=(0, 0, 0), param=0.0):
    m = np.zeros(1, dtype=oracle.MATERIAL_DTYPE)
    m["type"], m["albedo"], m["param"] = kind, albedo, param
    return m


def test_lambertian_scatter(oracle):
    m = _mat(oracle, 0, (0.1, 0.2, 0.3))
    u = [[0.25, 0.125]]
    ok, att, d = oracle.scatter_batch(m, [0], [[0, -1, 0]], [[0, 0, 0]], [[0, 1, 0]], [1], u)
    unit = oracle.unit_from_uniforms(u)[0]
    want = np.array([0, 1, 0]) + unit
    assert ok[0] and (att[0] == [0.1, 0.2, 0.3]).all() and np.allclose(d[0], want / np.linalg.norm(want), atol=1e-15)
    # degenerate: unit vector opposite to the normal -> the normal itself (src/material.rs:24-26)
    ok, att, d = oracle.scatter_batch(m, [0], [[0, -1, 0]], [[0, 0, 0]], [[0, 0, -1.0]], [1], [[0.0, 0.3]])  # u0 = 0 gives z = +1
    assert np.allclose(d[0], [0, 0, -1])


def test_metal_scatter_and_absorption(oracle):
    m = _mat(oracle, 1, (0.7, 0.6, 0.5), 0.0)
    v = np.array([1.0, -1.0, 0.0])
    ok, att, d = oracle.scatter_batch(m, [0], [v], [[0, 0, 0]], [[0, 1, 0]], [1], [[0.3, 0.3]])
    assert ok[0] and np.allclose(d[0], np.array([1, 1, 0]) / math.sqrt(2)) and (att[0] == [0.7, 0.6, 0.5]).all()
    # full fuzz with the unit vector pointing into the surface: absorbed (None, src/material.rs:36-38)
    m = _mat(oracle, 1, (0.7, 0.6, 0.5), 1.0)
    grazing = np.array([1.0, -1e-3, 0.0])
    ok, att, d = oracle.scatter_batch(m, [0], [grazing], [[0, 0, 0]], [[0, 1, 0]], [1], [[0.0, 0.0]])  # unit = (0,0,1): stays grazing above
    unit_down = [[0.5, 0.75]]  # z = 0, phi = 3pi/2 -> (0,-1,0)
    ok2, _, _ = oracle.scatter_batch(m, [0], [grazing], [[0, 0, 0]], [[0, 1, 0]], [1], unit_down)
    assert not ok2[0]


def test_dielectric_scatter(oracle):
    m = _mat(oracle, 2, param=1.5)
    # normal incidence, coin above the reflectance r0 = 0.04: refracts straight through
    ok, att, d = oracle.scatter_batch(m, [0], [[0, -1, 0]], [[0, 0, 0]], [[0, 1, 0]], [1], [[0.5, 0.0]])
    assert ok[0] and (att[0] == 1).all() and np.allclose(d[0], [0, -1, 0])
    # Schlick at cos = 1 is r0 = ((1-1.5)/(1+1.5))^2 = 0.04: a coin just below reflects, just above refracts
    ok, _, d = oracle.scatter_batch(m, [0, 0], [[0, -1, 0]] * 2, [[0, 0, 0]] * 2, [[0, 1, 0]] * 2, [1, 1], [[0.0399, 0], [0.0401, 0]])
    assert np.allclose(d[0], [0, 1, 0]) and np.allclose(d[1], [0, -1, 0])
    # leaving glass beyond the critical angle: total internal reflection whatever the coin (src/material.rs:57-59)
    v = np.array([math.sin(1.0), -math.cos(1.0), 0.0])
    ok, _, d = oracle.scatter_batch(m, [0], [v], [[0, 0, 0]], [[0, 1, 0]], [0], [[0.999999, 0]])
    assert np.allclose(d[0], [math.sin(1.0), math.cos(1.0), 0.0])


# ---- camera (src/camera.rs:40-82) ----
def test_camera_reference_sizes(oracle):
    for width, height in [(830, 466), (1920, 1080), (3840, 2160), (7680, 4320), (400, 225)]:
        assert oracle.camera_derive(oracle.make_config(image_width=width))[0] == height
    assert oracle.camera_derive(oracle.make_config(image_width=1, aspect_ratio=16.0))[0] == 1  # .max(1.0)


def test_camera_default_geometry(oracle):
    h, cam = oracle.camera_derive(oracle.make_config())
    # lookfrom -Z looking at the origin, vfov 90, focus 10: viewport 20 high at distance 10
    assert np.allclose(cam["center"], [0, 0, -1])
    assert np.allclose(cam["pixel_delta_v"], [0, -20.0 / 225, 0])
    assert np.allclose(np.abs(cam["pixel_delta_u"]), [20.0 * 400 / 225 / 400, 0, 0])
    assert (cam["defocus_disk_u"] == 0).all()


def test_quantise(oracle):
    q = oracle.quantise([[0.0, 1.0, 4.0], [0.25, -1.0, float("nan")], [0.999, 1e-9, 2.0]], 1)
    assert q.tolist() == [[0, 255, 255], [127, 0, 0], [254, 0, 255]]
    assert oracle.quantise([[40.0, 10.0, 2.5]], 10)[0].tolist() == [255, 255, 127]


# ---- golden vectors ----
def test_camera_fixture(oracle):
    z = np.load(os.path.join(GOLD, "camera.npz"))
    import raytracer_amd as rt

    h, cam = oracle.camera_derive(rt.config_rtiow().record().view(oracle.CONFIG_DTYPE))
    assert h == int(z["rtiow_height"]) == 466
    for k, v in cam.items():
        assert np.array_equal(v, z["rtiow_" + k])


def test_hits_fixture(oracle):
    z = np.load(os.path.join(GOLD, "hits_rtiow.npz"))
    _, _, scene = _scene(oracle)
    got = scene.trace_batch(z["org"], z["dir"], want_p=True)
    assert np.array_equal(got["prim"], z["prim"]) and np.array_equal(got["t"], z["t"])
    assert np.array_equal(got["normal"], z["normal"]) and np.array_equal(got["front"], z["front"]) and np.array_equal(got["p"], z["p"])
    assert np.array_equal(got["stats"], z["stats"])
    import raytracer_amd as rt

    org, d = oracle.primary_rays(rt.config_rtiow().record().view(oracle.CONFIG_DTYPE), 12345, z["ijs"])
    assert np.array_equal(org, z["org"][: len(org)]) and np.array_equal(d, z["dir"][: len(d)])


def test_scatter_fixture(oracle):
    z = np.load(os.path.join(GOLD, "scatter_rtiow.npz"))
    _, materials, _ = _scene(oracle)
    s, a, d = oracle.scatter_batch(materials.view(oracle.MATERIAL_DTYPE), z["material"], z["in_dir"], z["p"], z["normal"], z["front"], z["u"])
    assert np.array_equal(s, z["scattered"]) and np.array_equal(a, z["attenuation"]) and np.array_equal(d, z["out_dir"])
    assert (~s).sum() > 0 and s.sum() > 2000


def test_image_fixture(oracle):
    z = np.load(os.path.join(GOLD, "image_64x36_64spp.npz"))
    import raytracer_amd as rt

    _, _, scene = _scene(oracle)
    cfg = rt.config_rtiow(image_width=64, samples_per_pixel=64, max_depth=50).record().view(oracle.CONFIG_DTYPE)
    img, stats = scene.render(cfg, 12345, 0, 64, nthreads=os.cpu_count() or 1)
    assert np.array_equal(img, z["rgb_sum"]) and np.array_equal(stats, z["stats"])
    assert np.array_equal(oracle.quantise(img, 64), z["ppm_bytes"])
    # tiles and sample ranges compose exactly (each pixel's samples are added in index order)
    part = np.zeros_like(img)
    scene.render(cfg, 12345, 0, 64, tile=(0, 0, 64, 20), rgb_sum=part, nthreads=4)
    scene.render(cfg, 12345, 0, 64, tile=(0, 20, 64, 36), rgb_sum=part, nthreads=4)
    assert np.array_equal(part, img)


# ---- the direct sampling maps have the reference's distributions ----
def test_direct_maps_match_rejection_sampling_statistics(oracle):
    n = 200000
    rng = np.random.default_rng(1)
    direct = oracle.unit_from_uniforms(rng.random((n, 2)))
    rej = oracle.unit_rejection(99, n)
    assert np.abs(np.linalg.norm(direct, axis=1) - 1).max() < 1e-15 and np.abs(np.linalg.norm(rej, axis=1) - 1).max() < 1e-15
    for pts in (direct, rej):  # uniform on the sphere: zero mean, covariance I/3
        assert np.abs(pts.mean(axis=0)).max() < 5.0 / math.sqrt(3 * n)
        assert np.abs(np.cov(pts.T) - np.eye(3) / 3).max() < 5e-3
    # same distribution of each coordinate (two-sample KS statistic far below the 1e-3 critical value ~ 0.0062)
    for k in range(3):
        a, b = np.sort(direct[:, k]), np.sort(rej[:, k])
        grid = np.linspace(-1, 1, 401)
        ks = np.abs(np.searchsorted(a, grid) / n - np.searchsorted(b, grid) / n).max()
        assert ks < 0.0062
    disk_d = oracle.disk_from_uniforms(rng.random((n, 2)))
    disk_r = oracle.disk_rejection(7, n)
    for pts in (disk_d, disk_r):  # uniform in the disk: E r^2 = 1/2, zero mean
        assert abs((pts[:, :2] ** 2).sum(axis=1).mean() - 0.5) < 5e-3 and np.abs(pts.mean(axis=0)).max() < 5e-3
        assert (pts[:, 2] == 0).all() and ((pts ** 2).sum(axis=1) < 1).all()


def test_direct_and_rejection_renders_agree_within_noise(oracle):
    import raytracer_amd as rt

    _, _, scene = _scene(oracle)
    spp = 48
    cfg = rt.config_rtiow(image_width=48, samples_per_pixel=spp, max_depth=20).record().view(oracle.CONFIG_DTYPE)
    a, _ = scene.render(cfg, 1, 0, spp, mode=oracle.SAMPLER_DIRECT, nthreads=os.cpu_count() or 1)
    b, _ = scene.render(cfg, 2, 0, spp, mode=oracle.SAMPLER_DIRECT, nthreads=os.cpu_count() or 1)
    c, _ = scene.render(cfg, 3, 0, spp, mode=oracle.SAMPLER_REJECTION, nthreads=os.cpu_count() or 1)
    # the difference between the two samplers is no larger than between two seeds of one sampler
    noise = np.sqrt(((a - b) ** 2).mean())
    cross = np.sqrt(((a - c) ** 2).mean())
    assert cross < 1.25 * noise
    assert abs(a.mean() - c.mean()) < 0.02 * a.mean()


def test_camera_ray_draws_are_four_words_of_one_philox7_block(oracle):
    # The rule the render loop draws camera rays by, restated here from the generator's known-answer-checked words alone:
    # one Philox4x32-7 block with counter (pixel index, sample, 0, domain 0) and key = seed; uniform k = word k * 2^-32;
    # (0, 1) the position in the pixel, (2, 3) the point on the lens by the direct map r = sqrt(u2), phi = 2 pi u3
    # (src/camera.rs:114-136, src/vector.rs:29-40).  Pins the mapping so that oracle and kernels cannot drift together.
    import raytracer_amd as rt  # (host code only: the configuration record)

    cfg = rt.config_rtiow(830, 750, 50)
    ocfg = cfg.record().view(oracle.CONFIG_DTYPE)
    _, cam = oracle.camera_derive(ocfg)
    seed = 0x1234567855AA33CC
    rng = np.random.default_rng(17)
    ijs = np.stack([rng.integers(0, 830, 64), rng.integers(0, 466, 64), rng.integers(0, 750, 64)], axis=1).astype(np.uint32)
    org, d = oracle.primary_rays(ocfg, seed, ijs)
    for (i, j, s), o_got, d_got in zip(ijs, org, d):
        words = oracle.philox4x32([seed & 0xFFFFFFFF, seed >> 32], [int(j) * 830 + int(i), int(s), 0, 0], 7)
        u = [int(w) / 4294967296.0 for w in words]
        sample = cam["pixel00_loc"] + float(i) * cam["pixel_delta_u"] + float(j) * cam["pixel_delta_v"] \
            + ((-0.5 + u[0]) * cam["pixel_delta_u"] + (-0.5 + u[1]) * cam["pixel_delta_v"])
        r, phi = np.sqrt(u[2]), 2.0 * np.pi * u[3]
        o = cam["center"] + r * np.cos(phi) * cam["defocus_disk_u"] + r * np.sin(phi) * cam["defocus_disk_v"]
        want = (sample - o) / np.linalg.norm(sample - o)
        assert np.abs(o_got - o).max() <= 1e-13 and np.abs(d_got - want).max() <= 1e-13
    # a uniform is an exact multiple of 2^-32 below 1: the largest word gives 1 - 2^-32
    assert 0xFFFFFFFF / 4294967296.0 == 1.0 - 2.0 ** -32


def test_host_code_is_clean_under_the_sanitizers():
    # bvh_builder.cpp (threaded build, flatten, four-wide collapse), rt_host.cpp (camera, quantiser, PPM text, scenes) and the
    # oracle, compiled with -fsanitize=address,undefined and again with -fsanitize=thread, on scenes of 488 and 40 000 spheres
    # (tools/sanitize/run.sh; the run with 10^6 spheres is recorded in profiles/)
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["bash", os.path.join(root, "tools", "sanitize", "run.sh"), "40000"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert p.stdout.count("all checks passed") == 2 and "exit status 0" in p.stdout
