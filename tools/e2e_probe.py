#!/usr/bin/env python3
"""Wall time of the C++ driver (the reference's `cargo run -r > image.ppm`) from process start to the last byte of PPM text."""
import os
import subprocess
import sys
import time

exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "host", "raytracer")
for args in (["--width", "830", "--spp", "750"], ["--width", "830", "--spp", "750"], ["--width", "3840", "--spp", "1000"]):
    t0 = time.perf_counter()
    p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print("%s: wall %.3f s, %d bytes of PPM, stderr: %s" % (" ".join(args), dt, len(p.stdout), p.stderr.decode().strip().replace("\n", " | ")), flush=True)
