#!/usr/bin/env python3
"""Throughput of the render hot path on MI355X.

Metric: Mrays/s = closest-hit queries (primary + secondary) per second of the
render span (the span the reference times, src/camera.rs:140-156: scene
build, BVH build and PPM output excluded).

One step = one full render of the workload into a frame buffer resident in
HBM.  Default workload: the reference's scene (src/main.rs:42-96, 488
spheres) at 3840x2160, 1000 samples per pixel, depth 50.

With N > 1 ranks (one process per GPU, launched by torch.distributed.run)
rank r renders the sample range [r*spp, (r+1)*spp) of the whole frame and the
per-rank accumulation buffers are summed onto rank 0 with one RCCL reduce,
inside the timed region (weak scaling: per-GPU work is fixed).  With
--partition tiles the ranks render disjoint row bands of one frame instead and
nothing is exchanged (strong scaling).

On rank 0 at N = 1 the CPU oracle (a C++ f64 restatement of the reference's
Rayon loop; the reference itself cannot be built without a Rust toolchain)
is timed on a bounded sample of the 830x466 configuration on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X vector FP32 peak
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector FP64 peak
HBM_PEAK_GBS = 8000.0


def flops_per_ray(v, p):
    """Algorithmic work of one closest-hit query of the reference: 25 per node test (src/aabb.rs:73-78),
    28 per sphere test (src/sphere.rs:30-44), 60 for hit record + scatter + ray spawn; v and p are the
    reference traversal's node visits and sphere tests per ray (measured on the oracle)."""
    return 25.0 * v + 28.0 * p + 60.0


def host_threads():
    """Threads for the CPU baseline: the cores this process may use.  On a shared GPU host os.cpu_count() is the
    whole machine; the share of one GPU's job is what the scheduler exports in OMP_NUM_THREADS (or the affinity mask)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    return max(1, min(n, share) if share > 0 else n)


def cpu_baseline(spheres, materials, seconds_target=15.0):
    from oracle import oracle as orc

    import raytracer_amd as rt

    threads = host_threads()
    scene = orc.Scene(spheres.view(orc.SPHERE_DTYPE), materials.view(orc.MATERIAL_DTYPE))
    cfg = rt.config_rtiow(830, 750, 50).record().view(orc.CONFIG_DTYPE)
    # calibrate with two samples per pixel, then size the sample to the time target
    t0 = time.perf_counter()
    scene.render(cfg, 12345, 0, 2, nthreads=threads)
    dt = (time.perf_counter() - t0) / 2
    spp = int(max(1, min(748, seconds_target / max(dt, 1e-3))))
    t0 = time.perf_counter()
    _, st = scene.render(cfg, 12345, 2, 2 + spp, nthreads=threads)
    dt = time.perf_counter() - t0
    rays, visits, tests = (int(x) for x in st)
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "value_per_core": rays / dt / 1e6 / threads,
        "machine_cores": os.cpu_count(),
        "kind": "port",
        "build": "g++ -O3 -ffp-contract=off, one thread per core, pixels handed out dynamically",
        "sample": "830x466, %d of 750 spp, depth 50, %d rays in %.1f s; C++ f64 restatement of the Rayon loop, not Rayon" % (spp, rays, dt),
        "mean_path_length": rays / (830.0 * 466 * spp),
        "node_visits_per_ray": visits / rays,
        "sphere_tests_per_ray": tests / rays,
    }


def source_hash():
    """A hash of the sources the library is built from: counter profiles carry it (tools/publish_profiles.sh), and a
    profile whose hash differs from the tree's was measured on other kernels."""
    import glob
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "raytracer_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic():
    """HBM bytes per launch of the default workload from the newest committed counter profile (a separate rocprofv3
    --pmc run of this same command; counters cannot be collected from inside the timed run), or None.  `stale` says
    whether the profile was taken on other sources than the tree's."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            out = json.load(open(path))
        except (OSError, ValueError):
            continue
        out["file"] = "profiles/" + os.path.basename(path)
        out["stale"] = out.get("source_hash") != source_hash()
        return out
    return None


# The reference traversal's node visits and sphere tests per ray on each configuration (CPU oracle, 160-pixel-wide frame,
# 2 spp, render seed 12345: tools/run_configs.py --rooflines, profiles/r02_config_rooflines.jsonl).  Scenes and cameras
# are seeded, so these do not change from run to run; the 10^6-sphere oracle tree takes 78 s to build, too long for a bench run.
ORACLE_COUNTS = {
    "c1": (22.578, 1.522), "c2": (22.578, 1.522), "c5": (26.170, 1.505), "c3": (33.729, 1.460), "c4": (46.954, 1.326),
}
# name: (spheres, p_lambertian, p_metal, max_fuzz, width, spp, depth, scene seed) -- the sizes the baseline's configurations state
OTHER_CONFIGS = {
    "c1": (None, 0, 0, 0, 830, 750, 50, 1),
    "c5": (10000, 0.0, 0.5, 0.3, 3840, 128, 200, 5),
    "c3": (100000, 1 / 3, 1 / 3, 1.0, 3840, 64, 50, 3),
    "c4": (1000000, 0.15, 0.15, 1.0, 7680, 32, 50, 4),
}


def other_configs(rt, torch):
    """Three renders of each of the other configurations (outside the timed region) after a one-sample launch; the median is reported
    (these frames take 20 - 100 ms: a single launch right after scene creation reads up to 10 % low)."""
    out = []
    # (a process's first build on the device loads the builder's code, 12 ms once; build_ms below is what every later scene costs)
    ws, wm = rt.scene_field(1, 2000, 1 / 3, 1 / 3, 1.0)
    rt.BVH(ws, wm, builder=rt.BUILD_AUTO).close()
    for name, (n, pl, pm, fz, width, spp, depth, seed) in OTHER_CONFIGS.items():
        if n is None:
            spheres, materials = rt.scene_rtiow(seed)
            cfg = rt.config_rtiow(width, spp, depth)
        else:
            spheres, materials = rt.scene_field(seed, n, pl, pm, fz)
            cfg = rt.config_field(n, width, spp, depth)
        world = rt.BVH(spheres, materials, builder=rt.BUILD_AUTO)  # as rt_scene_create does: the reference's tree, built on the device from 1024 spheres on
        cam = cfg.camera()
        frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        cam.render_device(world, frame.data_ptr(), seed=1, spp_begin=0, spp_end=1)
        runs = []
        for _ in range(3):
            frame.zero_()
            torch.cuda.synchronize()
            runs.append(cam.render_device(world, frame.data_ptr(), seed=12345))
            torch.cuda.synchronize()
        st = sorted(runs, key=lambda r: r["kernel_ms"])[1]
        info = world.info()
        v, p = ORACLE_COUNTS[name]
        rate = st["rays"] / (st["kernel_ms"] * 1e-3)
        out.append({
            "name": name,
            "workload": "%d spheres, %dx%d, %d spp, max depth %d" % (len(spheres), width, cam.image_height, spp, depth),
            "Mrays_per_s": rate / 1e6, "kernel_ms": st["kernel_ms"], "kernel_ms_of_three": [r["kernel_ms"] for r in runs], "rays": st["rays"], "mean_path_length": st["rays"] / float(st["samples"]),
            "node_visits_per_ray": v, "sphere_tests_per_ray": p, "flops_per_ray": flops_per_ray(v, p),
            "frac": rate * flops_per_ray(v, p) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
            "build_ms": info["build_ms"], "builder": {0: "host", 1: "device, clustering", 3: "device, the reference's surface-area splits"}.get(int(info["builder"]), "?"), "walk": "two-wide tree in LDS" if info["n_nodes_in_lds"] == info["n_device_nodes"] else "four-wide tree, top in LDS",
            "frame_mean_radiance": float(frame.mean().item()) / spp,
        })
        world.close()
        del frame
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--partition", choices=["samples", "tiles"], default="samples")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the one render each of the other configurations after the timed region")
    ap.add_argument("--force-distributed", action="store_true", help="set up the process group and run the collectives even with one rank (a rehearsal of the N > 1 path on one GPU)")
    args = ap.parse_args()

    # Libraries write banners on the process's standard output (RCCL prints its version there when the process group
    # starts); the contract is one JSON line, so everything but that line goes to standard error.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd.parallel import partition_samples

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world_size, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    distributed = world_size > 1 or args.force_distributed
    if distributed:
        if world_size == 1:  # --force-distributed without a launcher: a process group of one
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
                os.environ.setdefault(k, v)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    rt.set_device(local_rank)  # the library's HIP runtime is not torch's: select the rank's GPU there too
    spheres, materials = rt.scene_rtiow(1)
    world = rt.BVH(spheres, materials)
    cfg = rt.config_rtiow(args.width, args.spp, args.depth)
    cam = cfg.camera()
    w, h = args.width, cam.image_height
    frame = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    # The library renders on its own runtime's default stream (stream=None): a handle of torch's runtime means nothing to it.
    # torch.cuda.synchronize() on both sides of a step orders the two.
    weak = args.partition == "samples"
    total_spp = args.spp * world_size if weak else args.spp
    last = {"rays": 0, "ms": 0.0, "launches": 0, "reduce_ms": 0.0}

    def render(s0, s1, tile):
        st = cam.render_device(world, frame.data_ptr(), seed=12345, spp_begin=s0, spp_end=s1, tile=tile)
        last["rays"] += st["rays"]
        last["ms"] += st["kernel_ms"]
        last["launches"] += 1

    def timed_reduce():  # the collective of the sample split, with its own clock
        torch.cuda.synchronize()
        t = time.perf_counter()
        dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        last["reduce_ms"] += 1e3 * (time.perf_counter() - t)

    def step():
        frame.zero_()
        torch.cuda.synchronize()  # torch's stream and the library's are not ordered with each other
        if weak:
            # as raytracer_amd.parallel.render_sample_split does, with the reduce under its own clock
            s0, s1 = partition_samples(0, total_spp, world_size)[rank]
            render(s0, s1, None)
            if distributed:
                timed_reduce()
        else:
            # one launch per rank: the rank's tile rows of the whole frame (dealt round-robin), as rt_render_multi does
            shard = cam.partition(world_size, rt.PARTITION_TILES, spp_begin=0, spp_end=total_spp)[rank]
            st = cam.render_shard_device(world, frame.data_ptr(), shard, seed=12345)
            last["rays"] += st["rays"]
            last["ms"] += st["kernel_ms"]
            last["launches"] += 1

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    last.update(rays=0, ms=0.0, launches=0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    rays, kernel_ms, launches = last["rays"], last["ms"], last["launches"]

    ranks = None
    if distributed:
        tot = torch.tensor([float(rays)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total_rays, elapsed_max = tot.item(), tmax.item()
        # what each rank ran on and how long its own kernels and reduces took: the first run on several GPUs must explain itself
        prop = torch.cuda.get_device_properties(local_rank)
        mine = {
            "rank": rank, "local_rank": local_rank, "hip_ordinal_torch": torch.cuda.current_device(), "hip_ordinal_library": rt.get_device(),
            "device": prop.name, "pci": "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0), getattr(prop, "pci_device_id", 0)),
            "visible_devices": torch.cuda.device_count(), "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"),
            "kernel_ms_per_step": kernel_ms / max(1, args.steps), "reduce_ms_per_step": last["reduce_ms"] / max(1, args.steps),
            "rays_per_step": rays / max(1, args.steps), "elapsed_s": elapsed,
        }
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, mine)
    else:
        total_rays, elapsed_max = float(rays), elapsed

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+secondary)",
            "value": total_rays / elapsed_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64 (sphere test, shading) + f32 (node test)",
            "data": "synthetic: seeded restatement of the reference scene (488 spheres), Philox seed 12345",
            "config": {
                "workload": "RTIOW-final scene of src/main.rs at %dx%d, %d spp per GPU, max depth %d"
                % (w, h, args.spp if weak else args.spp // max(1, world_size), args.depth),
                "partition": ("sample ranges per rank + one RCCL reduce" if weak else "row bands per rank, no exchange") if distributed else "single GPU",
                "rays_per_step": total_rays / args.steps,
                "mean_path_length": total_rays / args.steps / (float(w) * h * total_spp),
                "frame_mean_radiance": float(frame.mean().item()) / total_spp if rank == 0 else None,
            },
        }
        if ranks is not None:
            out["ranks_seen"] = dist.get_world_size()
            out["ranks"] = ranks
            out["distinct_devices"] = len({r["pci"] for r in ranks})
        v_bar, p_bar, source = 22.61, 1.52, "oracle counts of an earlier run (seed 12345, 830x466)"
        if world_size == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(spheres, materials)
            out["cpu_baseline"] = cb
            v_bar, p_bar, source = cb["node_visits_per_ray"], cb["sphere_tests_per_ray"], "cpu oracle sample of this run"
        if "cpu_baseline" in out:
            cb = out["cpu_baseline"]
            out["speedup_vs_cpu"] = {
                "vs_all_cores": out["value"] / cb["value"],
                "vs_one_core": out["value"] / cb["value_per_core"],
                "target_1000x_met": {"vs_all_cores": out["value"] / cb["value"] >= 1000.0, "vs_one_core": out["value"] / cb["value_per_core"] >= 1000.0},
            }
        k_ms = kernel_ms / max(1, launches)
        rays_per_launch = rays / max(1, launches)
        flops = rays_per_launch * flops_per_ray(v_bar, p_bar)
        # the node tests run in f32, the sphere tests and shading in f64: the time the work would take at both peaks
        f32_flops, f64_flops = rays_per_launch * 25.0 * v_bar, rays_per_launch * (28.0 * p_bar + 60.0)
        ideal_s = f32_flops / (FP32_VECTOR_PEAK_TFLOPS * 1e12) + f64_flops / (FP64_VECTOR_PEAK_TFLOPS * 1e12)
        traffic = measured_traffic() if (world_size == 1 and args.width == 3840 and args.spp == 1000 and args.depth == 50) else None
        out["roofline"] = {
            "bound": "valu",
            "kernel": "render_kernel",
            "achieved": flops / (k_ms * 1e-3) / 1e12,
            "peak": FP32_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": flops / (k_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
            "frac_meaning": "the reference's algorithmic flops per second over the FP32 vector peak",
            "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
            "traffic_replayed_from": traffic["file"] if traffic else None,
            "traffic_stale": traffic["stale"] if traffic else None,
            "source_hash": source_hash(),
            "kernel_ms": k_ms,
            "flops_per_ray": flops_per_ray(v_bar, p_bar),
            "per_ray_counts_from": source,
            "by_precision": {
                "f32_flops_per_ray": 25.0 * v_bar,
                "f64_flops_per_ray": 28.0 * p_bar + 60.0,
                "f32_peak": FP32_VECTOR_PEAK_TFLOPS,
                "f64_peak": FP64_VECTOR_PEAK_TFLOPS,
                "frac_of_mixed_ceiling": ideal_s / (k_ms * 1e-3),
            },
            "hbm_GBs_replayed": traffic["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 if traffic else None,
            "hbm_peak_GBs": HBM_PEAK_GBS,
            "note": "the whole scene (nodes, spheres, materials: 93 KB) lives in LDS, so neither HBM nor MFMA binds; "
                    "the kernel is bound by vector-ALU issue.  traffic and issue are not measured in this run: they are "
                    "replayed from the committed rocprofv3 --pmc profile of this same command (traffic_replayed_from; traffic_stale "
                    "says whether that profile was taken on the sources this run was built from)",
        }
        if traffic and "issue" in traffic and not traffic["stale"]:  # counter profile of the same workload and sources: how busy the vector units are
            issue = traffic["issue"]
            out["roofline"]["issue"] = {
                "valu_wave_instructions_per_launch": issue["valu_wave_instructions_per_launch"],
                "simd_cycles_per_valu_instruction": issue["simd_cycles_per_valu_instruction"],
                "mean_active_lanes_per_valu_instruction": issue["mean_active_lanes_per_valu_instruction"],
                "measured_cycles_per_instruction_alone": "2.5-2.7 (f32 add, mul, fma; 32-bit add, logic, mov), 4.1-4.6 (min/max, compare, select, shift, f64)",
                "replayed_from": traffic["file"] + ", profiles/r02_valu_rate.txt, profiles/r02_bank_rate.txt",
            }
        if world_size == 1 and not args.no_other_configs:
            head = {
                "name": "c2", "workload": "488 spheres, %dx%d, %d spp, max depth %d (the timed headline above)" % (w, h, args.spp, args.depth),
                "Mrays_per_s": rays_per_launch / (k_ms * 1e-3) / 1e6, "kernel_ms": k_ms, "rays": rays_per_launch,
                "mean_path_length": out["config"]["mean_path_length"], "node_visits_per_ray": v_bar, "sphere_tests_per_ray": p_bar,
                "flops_per_ray": flops_per_ray(v_bar, p_bar), "frac": out["roofline"]["frac"], "build_ms": world.info()["build_ms"], "builder": "host",
                "walk": "two-wide tree in LDS", "frame_mean_radiance": out["config"]["frame_mean_radiance"],
            }
            out["configs"] = [head] + other_configs(rt, torch)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.destroy_process_group()
    world.close()


if __name__ == "__main__":
    main()
