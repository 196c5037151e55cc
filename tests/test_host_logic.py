"""CPU-only checks of the shipped library's host side: the C ABI exports what
include/rt.h declares, the host BVH builder reproduces the oracle's (= the
reference's) tree node for node, the camera, the quantiser and the PPM text match
the oracle, and compute entry points fail loudly without a GPU."""
import os
import re
import tempfile

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(rt.EXPORTS), declared ^ set(rt.EXPORTS)
    lib = rt._lib.lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    spheres, materials = rt.scene_rtiow(1)
    try:
        world = rt.BVH(spheres, materials)
    except rt.RtError as e:
        assert e.code == 2 and "hip" in str(e).lower()  # RT_ERR_HIP: fails loudly, computes nothing
    else:
        world.close()
        pytest.skip("a GPU is present")


def test_scenes_beyond_the_documented_limits_are_refused():
    # one sphere more than rt_scene_create takes (include/rt.h): refused with RT_ERR_LIMIT before anything is read or built
    spheres = np.zeros(16777200, dtype=rt.SPHERE_DTYPE)
    materials = np.zeros(1, dtype=rt.MATERIAL_DTYPE)
    for builder in (rt.BUILD_HOST_SAH, rt.BUILD_DEVICE_LBVH, rt.BUILD_DEVICE_SAH, rt.BUILD_AUTO):
        with pytest.raises(rt.RtError) as e:
            rt.BVH(spheres, materials, builder=builder)
        assert e.value.code == 4 and "16777199" in str(e.value)
    with pytest.raises(rt.RtError) as e:
        rt.host_bvh(spheres, on_device=True)
    assert e.value.code == 4


def _same_tree(oracle, spheres):
    boxes, kind, soc, leaf, depth = rt.host_bvh(spheres)
    materials = np.zeros(1, dtype=oracle.MATERIAL_DTYPE)
    sp = spheres.copy()
    sp["material"] = 0
    ob, ok, os_, ol = oracle.Scene(sp.view(oracle.SPHERE_DTYPE), materials).bvh_dump()
    assert np.array_equal(boxes, ob) and np.array_equal(kind, ok) and np.array_equal(soc, os_) and np.array_equal(leaf, ol)
    return len(kind), depth


def test_host_bvh_equals_oracle_bvh(oracle):
    for seed in (1, 2, 3):
        n, depth = _same_tree(oracle, rt.scene_rtiow(seed)[0])
        assert n == 975 or n > 900
    n, depth = _same_tree(oracle, rt.scene_field(4, 3000, 0.15, 0.15)[0])
    assert depth < 40
    rng = np.random.default_rng(0)
    sp = np.zeros(400, dtype=rt.SPHERE_DTYPE)
    sp["center"] = rng.uniform(-50, 50, (400, 3))
    sp["radius"] = rng.uniform(0.01, 8, 400)  # heavily overlapping boxes
    _same_tree(oracle, sp)


def test_host_bvh_threaded_build_equals_oracle_bvh(oracle):
    # large enough for the builder to bin with several threads and to build subtrees on their own threads
    n, depth = _same_tree(oracle, rt.scene_field(3, 40000, 1.0 / 3.0, 1.0 / 3.0)[0])
    assert n > 70000


def test_host_bvh_edge_cases(oracle):
    sp = np.zeros(0, dtype=rt.SPHERE_DTYPE)
    boxes, kind, soc, leaf, depth = rt.host_bvh(sp)
    assert len(kind) == 1 and kind[0] == 3 and soc[0] == 0  # one empty leaf, as BVH::new builds (src/bvh.rs:108-110)
    one = np.zeros(1, dtype=rt.SPHERE_DTYPE)
    one["radius"] = 1.0
    assert _same_tree(oracle, one)[0] == 1
    same = np.zeros(5, dtype=rt.SPHERE_DTYPE)  # identical centroids: no valid axis, a single leaf
    same["radius"] = [1, 2, 3, 4, 5]
    assert _same_tree(oracle, same)[0] == 1
    line = np.zeros(64, dtype=rt.SPHERE_DTYPE)  # centroids on one axis only
    line["center"][:, 0] = np.arange(64)
    line["radius"] = 0.3
    n, depth = _same_tree(oracle, line)
    assert n == 127
    neg = np.zeros(3, dtype=rt.SPHERE_DTYPE)  # a negative radius bounds like a positive one (src/sphere.rs:50-52)
    neg["center"] = [[0, 0, 0], [5, 0, 0], [0, 7, 0]]
    neg["radius"] = [-1, 2, -0.5]
    _same_tree(oracle, neg)


def test_host_bvh_build_time_scales():
    import time

    sp = rt.scene_field(4, 200000, 0.15, 0.15)[0]
    t0 = time.time()
    boxes, kind, soc, leaf, depth = rt.host_bvh(sp)
    dt = time.time() - t0
    assert len(leaf) == len(sp) and sorted(leaf.tolist()) == list(range(len(sp)))
    assert dt < 10.0 and depth <= 64


def test_config_mirrors_reference_defaults_and_setters():
    c = rt.Config()
    assert c.get("aspect_ratio") == 16.0 / 9.0 and c.get("image_width") == 400 and c.get("samples_per_pixel") == 10
    assert c.get("max_depth") == 10 and c.get("vfov") == 90.0 and c.get("defocus_angle") == 0.0 and c.get("focus_dist") == 10.0
    assert c.get("lookfrom").tolist() == [0, 0, -1] and c.get("lookat").tolist() == [0, 0, 0] and c.get("vup").tolist() == [0, 1, 0]
    d = c.image_width(830).vfov(20.0)
    assert d.get("image_width") == 830 and c.get("image_width") == 400  # setters take and return by value (src/util.rs:12-16)


def test_camera_matches_oracle(oracle):
    cfgs = [rt.Config(), rt.config_rtiow(), rt.config_rtiow(3840, 1000, 50), rt.Config().image_width(7).aspect_ratio(0.37).vfov(120.0)
            .lookfrom((1.5, -2.0, 0.25)).lookat((0.1, 0.2, 0.3)).vup((0.1, 0.9, 0.2)).defocus_angle(3.0).focus_dist(2.5)]
    for cfg in cfgs:
        cam = cfg.camera()
        h, want = oracle.camera_derive(cfg.record().view(oracle.CONFIG_DTYPE))
        assert cam.image_height == h
        for k, v in want.items():
            assert np.array_equal(getattr(cam, k), v), k
    with pytest.raises(rt.RtError):
        rt.Config().image_width(0).camera()
    with pytest.raises(rt.RtError):
        rt.Config().lookfrom((float("nan"), 0, 0)).camera()


def test_quantise_and_ppm_match_oracle(oracle):
    rng = np.random.default_rng(4)
    w, h, spp = 37, 11, 7
    rgb = rng.uniform(-0.5, 9.0, (h, w, 3))
    rgb[0, 0] = [np.nan, np.inf, -np.inf]
    rgb[0, 1] = [0.0, spp * 1.0, spp * 0.25]
    assert np.array_equal(rt.quantise(rgb, spp), oracle.quantise(rgb, spp))
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "a.ppm"), os.path.join(d, "b.ppm")
        fd = os.open(a, os.O_WRONLY | os.O_CREAT)
        rt.write_ppm(fd, rgb, w, h, spp)
        os.close(fd)
        oracle.write_ppm(b, rgb, w, h, spp)
        text = open(a).read()
        assert text == open(b).read()
        lines = text.split("\n")
        assert lines[:3] == ["P3", "%d %d" % (w, h), "255"] and len(lines) == 3 + w * h + 1  # src/camera.rs:139,157-164
        assert lines[3] == "0 255 0" and lines[4] == "0 255 127"
    with pytest.raises(ValueError):
        rt.write_ppm(1, rgb, w + 1, h, spp)


def test_quantise_boundary_fixture_is_the_oracles_and_the_hosts(oracle):
    # the committed byte-boundary vector (tests/golden/make_c1_fixture.py): the oracle still produces it, and so does the host quantiser
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quantise_boundaries.npz"))
    sums, spp = z["sums"].reshape(-1, 3), int(z["spp"])
    assert np.array_equal(oracle.quantise(sums, spp), z["bytes"])
    assert np.array_equal(rt.quantise(sums, spp), z["bytes"])


def test_rtiow_scene_follows_main_rs():
    spheres, materials = rt.scene_rtiow(1)
    assert len(spheres) == 488 and len(materials) == 488  # 4 fixed + 22 x 22 (src/main.rs:45-82)
    assert spheres["center"][:4].tolist() == [[0, -1000, 0], [0, 1, 0], [-4, 1, 0], [4, 1, 0]]
    assert spheres["radius"][:4].tolist() == [1000, 1, 1, 1] and (spheres["radius"][4:] == 0.2).all()
    assert materials["type"][:4].tolist() == [0, 2, 0, 1] and materials["param"][1] == 1.5 and materials["param"][3] == 0.0
    assert materials["albedo"][2].tolist() == [0.4, 0.2, 0.1] and materials["albedo"][3].tolist() == [0.7, 0.6, 0.5]
    grid = spheres["center"][4:].reshape(22, 22, 3)
    a = np.arange(-11, 11)
    assert ((grid[:, :, 0] >= a[:, None]) & (grid[:, :, 0] < a[:, None] + 0.9)).all()  # a + 0.9 * U[0,1)
    assert ((grid[:, :, 2] >= a[None, :]) & (grid[:, :, 2] < a[None, :] + 0.9)).all() and (grid[:, :, 1] == 0.2).all()
    kinds = np.bincount(materials["type"][4:], minlength=3)
    assert (kinds > 120).all()  # uniform thirds (src/main.rs:64)
    metal = materials[4:][materials["type"][4:] == 1]
    assert (metal["albedo"] >= 0.5).all() and (metal["albedo"] <= 1.0).all() and (metal["param"] >= 0).all() and (metal["param"] < 1).all()
    lam = materials[4:][materials["type"][4:] == 0]
    assert (lam["albedo"] >= 0).all() and (lam["albedo"] < 1).all() and abs(lam["albedo"].mean() - 0.25) < 0.03  # product of two uniforms
    assert (materials["param"][4:][materials["type"][4:] == 2] == 1.5).all()
    again = rt.scene_rtiow(1)
    assert np.array_equal(again[0], spheres) and not np.array_equal(rt.scene_rtiow(2)[0], spheres)


def test_flatten_shares_materials_by_identity():
    glass = rt.Dielectric().ir(1.5)
    objs = [rt.Sphere((0, 0, 0), 1.0, glass), rt.Sphere((3, 0, 0), 1.0, glass), rt.Sphere((6, 0, 0), 1.0, rt.Metal((0.5, 0.5, 0.5)).fuzz(0.1))]
    spheres, materials = rt.flatten(objs)
    assert spheres["material"].tolist() == [0, 0, 1] and materials["type"].tolist() == [2, 1] and materials["param"].tolist() == [1.5, 0.1]


def test_cxx_driver_fails_loudly_without_a_gpu():
    import subprocess

    exe = os.path.join(ROOT, "raytracer_amd", "host", "raytracer")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    p = subprocess.run([exe, "--width", "32", "--spp", "1"], capture_output=True, timeout=60)
    assert p.returncode == 1 and p.stdout == b"" and b"hip" in p.stderr.lower()


def test_debug_knobs_refuse_values_that_would_hang_a_wave():
    import ctypes as C

    L = rt._lib.lib()
    for name in (b"inner_min", b"leaf_min"):
        for bad in (0.0, -1.0, 65.0, 2.5, float("nan"), float("inf")):
            assert L.rt_debug_set(name, C.c_double(bad)) == 1, (name, bad)
            assert b"1 .. 64" in L.rt_last_error()
        assert L.rt_debug_set(name, C.c_double(10.0 if name == b"inner_min" else 8.0)) == 0  # (the defaults)
    for bad in (-1.0, 65.0, float("nan")):
        assert L.rt_debug_set(b"both_min", C.c_double(bad)) == 1
    assert L.rt_debug_set(b"both_min", C.c_double(0.0)) == 0 and L.rt_debug_set(b"both_min", C.c_double(8.0)) == 0
    assert L.rt_debug_set(b"batches_per_wave", C.c_double(0.0)) == 1 and L.rt_debug_set(b"batches_per_wave", C.c_double(64.0)) == 0
    assert L.rt_debug_set(b"no_such_knob", C.c_double(1.0)) == 1


def test_partition_covers_the_request_exactly_once():
    # what rt_render_multi hands to each device (host code only)
    cam = rt.config_rtiow(image_width=131, samples_per_pixel=50).camera()
    h = cam.image_height
    for tile in (None, (3, 5, 120, 70), (0, 0, 131, 7), (10, 10, 10, 40)):
        x0, y0, x1, y1 = tile if tile else (0, 0, 131, h)
        for n in (1, 2, 3, 8, 16):
            shards = cam.partition(n, rt.PARTITION_TILES, spp_begin=4, spp_end=50, tile=tile)
            owner = np.full(max(0, y1 - y0), -1)
            for g, s in enumerate(shards):
                assert (int(s["x0"]), int(s["y0"]), int(s["x1"]), int(s["y1"])) == (x0, y0, x1, y1)
                assert (int(s["spp_begin"]), int(s["spp_end"])) == (4, 50) and int(s["row_stride"]) == n and int(s["row_first"]) == g
                rows = [r for r in range(y1 - y0) if (r // 8) % n == g]
                assert (owner[rows] == -1).all()
                owner[rows] = g
            assert (owner >= 0).all()
            shards = cam.partition(n, rt.PARTITION_SAMPLES, spp_begin=4, spp_end=50, tile=tile)
            at, sizes = 4, []
            for s in shards:
                assert int(s["row_first"]) == 0 and int(s["row_stride"]) == 1 and int(s["spp_begin"]) == at
                sizes.append(int(s["spp_end"]) - at)
                at = int(s["spp_end"])
            assert at == 50 and max(sizes) - min(sizes) <= 1
    with pytest.raises(rt.RtError):
        cam.partition(0)
    with pytest.raises(rt.RtError):
        cam.partition(2, tile=(0, 0, 200, 10))
    with pytest.raises(rt.RtError):
        cam.partition(2, partition=2)


def _c_declarations(text):
    """{name: [parameter types]} for the functions and {name: [(type, field)]} for the structs of a C header."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\} (\w+);", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.rsplit(" ", 1)[0], decl.split(" ", 1)[1] if decl.count(" ") == 1 else None
            m = re.match(r"(\w+)\s+(.*)", decl)
            ctype, names = m.group(1), m.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                dim = re.search(r"\[(\d+)\]", nm)
                fields.append((ctype + ("[%s]" % dim.group(1) if dim else ""), re.sub(r"\[.*", "", nm)))
        structs[name] = fields
    funcs = {}
    for ret, name, params in re.findall(r"^(int|void|const char\*)\s+(rt_\w+)\(([^)]*)\);", text, flags=re.M | re.S):
        plist = []
        for p in params.split(","):
            p = " ".join(p.split())
            if p in ("void", ""):
                continue
            ptype = re.sub(r"\s*\w+$", "", p)  # drop the parameter name
            plist.append(ptype.replace(" *", "*"))
        funcs[name] = plist
    return structs, funcs


_RUST_TYPES = {
    "u8": "uint8_t", "u32": "uint32_t", "u64": "uint64_t", "i32": "int32_t", "f32": "float", "f64": "double", "usize": "size_t",
    "c_int": "int", "c_char": "char", "c_void": "void",
}


def _rust_type_to_c(t):
    t = t.strip()
    m = re.match(r"\*(const|mut) (.*)", t)
    if m:
        inner = _rust_type_to_c(m.group(2))
        return ("const " if m.group(1) == "const" else "") + inner + "*"
    m = re.match(r"\[(\w+); (\d+)\]", t)
    if m:
        return "%s[%s]" % (_RUST_TYPES[m.group(1)], m.group(2))
    return _RUST_TYPES.get(t, t)


def test_rust_ffi_declarations_match_the_header():
    # No Rust toolchain exists here, so the crate under rust/ cannot be compiled; this keeps its raw declarations
    # honest: every extern function and #[repr(C)] struct in rust/src/ffi.rs must agree with include/rt.h in name,
    # argument count, argument types and field order.
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs, funcs = _c_declarations(open(os.path.join(root, "include", "rt.h")).read())
    src = open(os.path.join(root, "rust", "src", "ffi.rs")).read()
    src = re.sub(r"//.*", "", src)
    n_structs = 0
    for name, body in re.findall(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", src, flags=re.S):
        if name == "rt_scene":
            continue  # opaque
        fields = [(f.strip().split(":")[0].replace("pub ", "").strip(), _rust_type_to_c(f.split(":")[1])) for f in body.split(",") if ":" in f]
        want = structs[name]
        assert len(fields) == len(want), name
        for (rname, rtype), (ctype, cname) in zip(fields, want):
            assert rtype == ctype, (name, rname, rtype, ctype)
            assert rname == cname or (rname, cname) == ("kind", "type"), (name, rname, cname)  # `type` is a Rust keyword
        n_structs += 1
    assert n_structs >= 5
    block = re.search(r'extern "C" \{(.*)\n\}', src, flags=re.S).group(1)
    n_funcs = 0
    for name, params, ret in re.findall(r"pub fn (rt_\w+)\((.*?)\)\s*->\s*([^;]+);", block, flags=re.S):
        assert name in funcs, name + " is not declared in include/rt.h"
        rparams = [_rust_type_to_c(p.split(":", 1)[1]) for p in params.split(",") if ":" in p]
        assert rparams == funcs[name], (name, rparams, funcs[name])
        n_funcs += 1
    assert n_funcs >= 10
    # the C constants the crate repeats
    header = open(os.path.join(root, "include", "rt.h")).read()
    for cname, value in re.findall(r"pub const (RT_\w+): u32 = (\d+);", src):
        assert re.search(r"\b%s = %s\b" % (cname, value), header), cname


# ---- the scene-sized leaf taken out of the device's copy of the tree (host logic of scene creation) ----
DEV_NODE = np.dtype([("x", "<f4", 4), ("y", "<f4", 4), ("z", "<f4", 4), ("ref0", "<u4"), ("ref1", "<u4"), ("axis", "<u4"), ("pad", "<u4")])
LEAF, NONE = 0x80000000, 0xFFFFFFFF


def _leaf(slot, count=1):
    return LEAF | (count << 24) | slot


def _node(box0, ref0, box1, ref1):
    n = np.zeros((), dtype=DEV_NODE)
    for axis, name in enumerate("xyz"):
        n[name] = (box0[0][axis], box0[1][axis], box1[0][axis], box1[1][axis])
    n["ref0"], n["ref1"] = ref0, ref1
    return n


def _take_out(nodes, root=0):
    import ctypes as C

    nodes = np.ascontiguousarray(nodes, dtype=DEV_NODE)
    out = np.zeros_like(nodes)
    first, start = C.c_uint32(), C.c_uint32()
    rc = rt._lib.lib().rt_debug_take_out_first_leaf(nodes.ctypes.data_as(C.c_void_p), C.c_size_t(len(nodes)), C.c_uint32(root),
                                                    out.ctypes.data_as(C.c_void_p), C.byref(first), C.byref(start))
    assert rc == 0
    return out, first.value, start.value


def _child_box(n, c):
    return tuple(float(n[a][2 * c]) for a in "xyz"), tuple(float(n[a][2 * c + 1]) for a in "xyz")


BIG = ((-1000.0, -2000.0, -1000.0), (1000.0, 0.0, 1000.0))  # the box of a ground sphere
A, B, C_, D = ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), ((2.0, 0.0, 0.0), (3.0, 1.0, 1.0)), ((5.0, 0.0, 4.0), (6.0, 1.0, 5.0)), ((8.0, 0.0, 0.0), (9.0, 2.0, 1.0))


def _union(p, q):
    return tuple(min(a, b) for a, b in zip(p[0], q[0])), tuple(max(a, b) for a, b in zip(p[1], q[1]))


def test_take_out_leaf_directly_under_the_root():
    # root = (ground leaf, node 1); node 1 = (leaf, leaf): no node changes, the walk starts at node 1
    nodes = np.array([_node(BIG, _leaf(7), _union(A, B), 1), _node(A, _leaf(0), B, _leaf(1))])
    out, first, start = _take_out(nodes)
    assert first == 7 and start == 1 and out.tobytes() == nodes.tobytes()
    # the same with the ground as child 1 and a leaf as its sibling: the walk starts at that leaf
    nodes = np.array([_node(A, _leaf(3), BIG, _leaf(9))])
    out, first, start = _take_out(nodes)
    assert first == 9 and start == _leaf(3) and out.tobytes() == nodes.tobytes()


def test_take_out_leaf_two_levels_down_shrinks_the_ancestors():
    # root 0 = (node 1, node 2); node 1 = (node 3, leaf D); node 3 = (ground, node 4); node 4 = (A, B); node 2 = (C, C)
    below3 = _union(BIG, _union(A, B))
    below1 = _union(below3, D)
    nodes = np.array([_node(below1, 1, C_, 2), _node(below3, 3, D, _leaf(5)), _node(C_, _leaf(6), C_, _leaf(6)),
                      _node(BIG, _leaf(4), _union(A, B), 4), _node(A, _leaf(0), B, _leaf(1))])
    out, first, start = _take_out(nodes)
    assert first == 4 and start == 0
    # node 1's slot for node 3 now holds node 4 with node 4's box; the root's slot for node 1 holds what is left below node 1
    assert out[1]["ref0"] == 4 and _child_box(out[1], 0) == _union(A, B) and out[1]["ref1"] == _leaf(5) and _child_box(out[1], 1) == D
    assert out[0]["ref0"] == 1 and _child_box(out[0], 0) == _union(_union(A, B), D) and _child_box(out[0], 1) == C_
    assert out[2:].tobytes() == nodes[2:].tobytes()  # nothing else is touched (node 3 is no longer reachable)


def test_take_out_finds_nothing_to_take():
    small = np.array([_node(A, _leaf(0), B, _leaf(1))])
    assert _take_out(small)[1:] == (NONE, 0)  # no leaf with half the tree's area
    two = np.array([_node(BIG, _leaf(0, count=2), A, _leaf(2))])
    assert _take_out(two)[1:] == (NONE, 0)  # a leaf of two spheres stays
    # too deep: the ground hangs below the fourth level
    deep = [_node(_union(BIG, A), k + 1, D, _leaf(20 + k)) for k in range(4)] + [_node(BIG, _leaf(9), A, _leaf(0))]
    out, first, start = _take_out(np.array(deep))
    assert (first, start) == (NONE, 0) and out.tobytes() == np.array(deep).tobytes()
    assert _take_out(np.zeros(0, dtype=DEV_NODE), root=_leaf(0))[1:] == (NONE, _leaf(0))  # the tree is a single leaf
