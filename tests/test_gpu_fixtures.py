"""The HIP path against the committed golden vectors (tests/golden/*.npz,
written by make_fixtures.py), through the C ABI.  These run without building
the oracle."""
import os

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def world():
    z = np.load(os.path.join(GOLD, "scene_rtiow_seed1.npz"))
    spheres, materials = rt.scene_rtiow(1)
    assert np.array_equal(spheres, z["spheres"]) and np.array_equal(materials, z["materials"])
    w = rt.BVH(spheres, materials)
    yield w
    w.close()


def test_camera_fixture():
    z = np.load(os.path.join(GOLD, "camera.npz"))
    for prefix, cfg in (("rtiow_", rt.config_rtiow()), ("default_", rt.Config())):
        cam = cfg.camera()
        assert cam.image_height == int(z[prefix + "height"])
        for k in ("center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v", "defocus_disk_u", "defocus_disk_v"):
            assert np.array_equal(getattr(cam, k), z[prefix + k])


def test_hits_fixture(world):
    z = np.load(os.path.join(GOLD, "hits_rtiow.npz"))
    got = world.trace_batch(z["org"], z["dir"])
    assert np.array_equal(got["prim"], z["prim"])  # closest-hit sphere index: exact
    hit = z["prim"] >= 0
    assert np.isinf(got["t"][~hit]).all()
    assert (np.abs(got["t"][hit] - z["t"][hit]) <= 1e-5 * np.abs(z["t"][hit])).all()  # bar: 1e-5 relative
    assert np.abs(got["t"][hit] - z["t"][hit]).max() / np.abs(z["t"][hit]).max() < 1e-9  # observed: ~1e-12
    assert np.abs(got["normal"] - z["normal"]).max() <= 1e-5 and np.array_equal(got["front"], z["front"])
    org, d = rt.config_rtiow().camera().primary_rays(z["ijs"], seed=12345)
    n = len(z["ijs"])
    assert np.abs(org - z["org"][:n]).max() <= 1e-12 and np.abs(d - z["dir"][:n]).max() <= 1e-12


def test_scatter_fixture(world):
    z = np.load(os.path.join(GOLD, "scatter_rtiow.npz"))
    s, a, d = world.scatter_batch(z["material"], z["in_dir"], z["p"], z["normal"], z["front"], z["u"])
    assert np.array_equal(s, z["scattered"])
    assert np.abs(a - z["attenuation"]).max() <= 1e-12 and np.abs(d - z["out_dir"]).max() <= 1e-9


def test_image_fixture(world):
    z = np.load(os.path.join(GOLD, "image_64x36_64spp.npz"))
    cam = rt.config_rtiow(image_width=64, samples_per_pixel=64, max_depth=50).camera()
    got, stats = cam.render_sum(world, seed=12345)
    want = z["rgb_sum"]
    close = np.abs(got - want).max(axis=2) <= 1e-9 * np.maximum(1.0, np.abs(want).max(axis=2))
    print("image fixture: %.2f%% of pixels equal to 1e-9" % (100 * close.mean()))
    assert close.mean() >= 0.995  # a last-bit difference can redirect a deep path; almost every pixel agrees to 1e-9
    assert abs(stats["rays"] - int(z["stats"][0])) <= 0.002 * int(z["stats"][0])
    q = rt.quantise(got, 64)
    diff = q.astype(int) - z["ppm_bytes"].astype(int)
    assert (diff != 0).any(axis=1).mean() <= 0.03
    mse = (diff.astype(float) ** 2).mean()
    assert mse == 0 or 10 * np.log10(255.0 ** 2 / mse) >= 45.0


def test_reference_configuration_whole_frame(world):
    # src/main.rs:86-96 as it stands -- 830 x 466, 750 samples per pixel, depth 50 -- against the oracle's render of the
    # whole frame (tests/golden/make_c1_fixture.py): the PPM bytes of every pixel and the f64 sums of every 8th row
    z = np.load(os.path.join(GOLD, "image_c1_830x466_750spp.npz"))
    cam = rt.config_rtiow(image_width=830, samples_per_pixel=750, max_depth=50).camera()
    assert cam.image_height == 466
    got, stats = cam.render_sum(world, seed=12345)
    want_rows = z["rgb_sum_rows"]
    rows = z["rows"]
    close = np.abs(got[rows] - want_rows).max(axis=2) <= 1e-9 * np.maximum(1.0, np.abs(want_rows).max(axis=2))
    q = rt.quantise(got, 750)
    same_bytes = (q == z["ppm_bytes"]).all(axis=1)
    o_rays = int(z["stats"][0])
    print("830x466x750: %.3f%% of pixels (every 8th row) equal to 1e-9, %.3f%% of PPM pixels byte-equal (worst byte step %d), rays %d vs oracle %d"
          % (100 * close.mean(), 100 * same_bytes.mean(), np.abs(q.astype(int) - z["ppm_bytes"].astype(int)).max(), stats["rays"], o_rays))
    assert close.mean() >= 0.999 and same_bytes.mean() >= 0.999
    assert np.abs(q.astype(int) - z["ppm_bytes"].astype(int)).max() <= 1
    assert abs(stats["rays"] - o_rays) <= 0.001 * o_rays and stats["samples"] == 830 * 466 * 750


def test_device_quantise_equals_oracle_on_the_byte_boundaries():
    # write_colour (src/camera.rs:85-95) on the device against bytes the oracle produced (tests/golden/make_c1_fixture.py): sums
    # whose gamma-corrected value sits on, just below and just above every byte boundary, NaN, infinities, negatives
    import torch

    z = np.load(os.path.join(GOLD, "quantise_boundaries.npz"))
    sums, spp = z["sums"], int(z["spp"])
    dev = torch.from_numpy(sums).cuda()
    out = torch.zeros(len(sums) + 2, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rt.quantise_device(dev.data_ptr(), len(sums) // 3, spp, out.data_ptr())
    got = out[: len(sums)].cpu().numpy().reshape(-1, 3)
    assert np.array_equal(got, z["bytes"])
    assert len({int(b) for b in got.reshape(-1)}) == 256  # every byte value occurs


def test_cxx_driver_prints_the_same_ppm(world, tmp_path):
    # the C++ mirror of src/main.rs (host/main.cpp) against the ctypes path and the committed image
    import subprocess

    exe = os.path.join(os.path.dirname(GOLD), "..", "raytracer_amd", "host", "raytracer")
    p = subprocess.run([exe, "--width", "64", "--spp", "64", "--depth", "50", "--seed", "1", "--render-seed", "12345"],
                       capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stderr.startswith(b"Completed in ") and p.stderr.strip().endswith(b"seconds")  # src/camera.rs:156
    cam = rt.config_rtiow(image_width=64, samples_per_pixel=64, max_depth=50).camera()
    got, _ = cam.render_sum(world, seed=12345)
    path = tmp_path / "py.ppm"
    fd = os.open(path, os.O_WRONLY | os.O_CREAT)
    rt.write_ppm(fd, got, 64, cam.image_height, 64)
    os.close(fd)
    assert p.stdout == open(path, "rb").read()
    lines = p.stdout.decode().split("\n")
    assert lines[:3] == ["P3", "64 36", "255"]
    px = np.array([[int(v) for v in ln.split()] for ln in lines[3:3 + 64 * 36]])
    z = np.load(os.path.join(GOLD, "image_64x36_64spp.npz"))
    assert (px != z["ppm_bytes"].astype(int)).any(axis=1).mean() <= 0.03


def test_device_quantise_equals_host_quantise():
    # write_colour (src/camera.rs:85-95) on the device must give the host's bytes, including at the
    # values where 255*sqrt(x/spp) crosses an integer and for NaN, infinities and negatives
    import torch

    rng = np.random.default_rng(8)
    spp = 7
    k = np.arange(0, 256, dtype=np.float64)
    edge = (k / 255.0) ** 2 * spp  # sums whose gamma-corrected value sits on a byte boundary
    edges = np.concatenate([np.nextafter(edge, -np.inf), edge, np.nextafter(edge, np.inf),
                            np.nextafter(np.nextafter(edge, np.inf), np.inf), np.nextafter(np.nextafter(edge, -np.inf), -np.inf)])
    special = np.array([np.nan, np.inf, -np.inf, -1.0, -0.0, 0.0, 1e-320, 1e-300, 1e300, spp * 1.0, spp * (1 - 2 ** -53)])
    vals = np.concatenate([edges, special, rng.uniform(0, spp, 3_000_000), rng.uniform(0, 3 * spp, 200_000), 10.0 ** rng.uniform(-12, 2, 500_000)])
    vals = vals[: len(vals) // 3 * 3].copy()
    n_px = len(vals) // 3
    want = rt.quantise(vals.reshape(-1, 3), spp)
    dev = torch.from_numpy(vals).cuda()
    out = torch.zeros(3 * n_px + 2, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # torch's stream and the library's are not ordered with each other
    rt.quantise_device(dev.data_ptr(), n_px, spp, out.data_ptr())
    got = out[: 3 * n_px].cpu().numpy().reshape(-1, 3)
    assert np.array_equal(got, want)
    odd = vals[:5 * 3 + 0][: 15].copy()  # an odd number of components goes through the tail path
    dev = torch.from_numpy(odd).cuda()
    out = torch.zeros(16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rt.quantise_device(dev.data_ptr(), 5, spp, out.data_ptr())
    assert np.array_equal(out[:15].cpu().numpy().reshape(-1, 3), rt.quantise(odd.reshape(-1, 3), spp))


def test_render_ppm_equals_render_then_write(world, tmp_path):
    cfg = rt.config_rtiow(image_width=96, samples_per_pixel=8, max_depth=50)
    cam = cfg.camera()
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    fd = os.open(a, os.O_WRONLY | os.O_CREAT)
    stats = cam.render(world, seed=3, fd=fd)
    os.close(fd)
    got, st2 = cam.render_sum(world, seed=3)
    fd = os.open(b, os.O_WRONLY | os.O_CREAT)
    rt.write_ppm(fd, got, 96, cam.image_height, 8)
    os.close(fd)
    assert open(a, "rb").read() == open(b, "rb").read() and stats["rays"] == st2["rays"]


def test_converged_image_matches_the_reference_samplers(world):
    # The GPU path draws unit vectors and disk points by direct maps; the reference uses rejection loops
    # (src/vector.rs:16-47).  Same distributions, so converged images must agree: RMSE on linear radiance, the
    # mean (bias) and PSNR on the 8-bit values, against a 4096-spp image made with the reference's samplers.
    z = np.load(os.path.join(GOLD, "image_64x36_4096spp_reference_samplers.npz"))
    cam = rt.config_rtiow(image_width=64, samples_per_pixel=4096, max_depth=50).camera()
    got, stats = cam.render_sum(world, seed=777)
    got /= 4096.0
    want = z["rgb_mean"]
    rmse = np.sqrt(((got - want) ** 2).mean())
    bias = (got - want).mean()
    qa, qb = rt.quantise(got, 1).astype(float), rt.quantise(want, 1).astype(float)
    psnr = 10 * np.log10(255.0 ** 2 / ((qa - qb) ** 2).mean())
    print("converged 64x36x4096: rmse %.4f bias %.1e psnr %.1f dB" % (rmse, bias, psnr))
    assert rmse <= 0.006 and abs(bias) <= 2e-4 and psnr >= 45.0  # two independent 4096-spp estimates differ by 0.003 / 50 dB
    assert abs(stats["rays"] / stats["samples"] - float(z["stats"][0]) / (64 * 36 * 4096)) <= 0.01  # mean path length
