#!/usr/bin/env python3
"""Writes tests/golden/reference_image_rows.npz: the top 224 rows (RGB, 8 bits) of the example render the reference
keeps in its repository (image.png, 830 x 466, shown in its README.md:4), the only output of the reference itself
that exists.  Its scene was random (src/main.rs:57-82 draws from an unseeded generator), but two regions of the
image do not depend on the random spheres: the sky above the horizon, which is a function of the camera, the sky
gradient and write_colour alone (src/camera.rs:40-82, 85-95, 103-107, 114-136), and the ground far outside the
field of small spheres, whose mean colour is a function of the ground's sphere test, the hit record, the Lambertian
scatter and the recursion of ray_colour (src/sphere.rs:29-48, src/hit.rs:18-31, src/material.rs:21-30,
src/camera.rs:98-112).  Two more follow from the three fixed spheres (src/main.rs:49-55): the part of the large metal
sphere that mirrors only sky, and the part of the large glass sphere through which only sky is seen (rows 149-215).
tests/test_reference_image.py compares the oracle and the GPU path with these rows.

The PNG is decoded here (zlib + the five PNG row filters); no image library is needed.  Re-run where the
reference is present:   python tests/golden/make_reference_image_fixture.py /root/reference/image.png
"""
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = 224


def read_png(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat = 8, b""
    while pos < len(d):
        n, typ = struct.unpack(">I4s", d[pos:pos + 8])
        body = d[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            w, h, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
    assert depth == 8 and interlace == 0 and colour in (2, 6)
    ch = 3 if colour == 2 else 4
    raw, stride = zlib.decompress(idat), w * ch
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = [0] * stride
    p = 0
    for y in range(h):
        ft, line = raw[p], raw[p + 1:p + 1 + stride]
        p += 1 + stride
        cur = [0] * stride
        for i in range(stride):
            a = cur[i - ch] if i >= ch else 0
            b = prev[i]
            c = prev[i - ch] if i >= ch else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
        if y + 1 >= ROWS:
            break
    return out.reshape(h, w, ch), w, h


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/image.png"
    img, w, h = read_png(path)
    np.savez_compressed(os.path.join(HERE, "reference_image_rows.npz"), rgb=img[:ROWS, :, :3].copy(), width=w, height=h,
                        source="image.png of the reference repository (README.md:4), rows 0..%d" % (ROWS - 1))
    print("wrote", ROWS, "rows of", w, "x", h)


if __name__ == "__main__":
    main()
