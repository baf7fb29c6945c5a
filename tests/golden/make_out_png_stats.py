#!/usr/bin/env python3
"""The one image the reference's own binary rendered of its TEXTURE path: raytracer/output/out.png (800x600: the earth,
the moon inside a glass ball, a mirror ground).  It is not reproducible from the committed JSON (other camera, other
`h_offset`, and it predates light sampling) — but which way is UP on the textured sphere
(`v = n.y * 0.5 + 0.5`, the row flip of materials.rs:243) and which way is EAST (`u = atan2(n.x, n.z) / 2pi + 0.5`,
sphere.rs:35-43) do not depend on any of that.

    python tests/golden/make_out_png_stats.py      # here (reads /root/reference, which the GPU box does not have)

finds the earth's disc (least-squares circle through its silhouette) and writes tests/golden/out_png_earth.npz: the disc
resampled on a 96 x 96 grid of VIEW-space coordinates (x right, y up, both in units of the disc's radius): the mean RGB8
colour of every cell (NaN outside the disc).  tests/test_texture_orientation.py rebuilds a
latitude-longitude map from it (for a few assumed camera elevations) and slides it over the map of OUR renders."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/raytracer/output/out.png"
GRID = 96


def find_disc(im):
    lum = im.sum(axis=2)
    m = lum > 45
    m[:, :300] = False; m[:, 500:] = False; m[:195] = False; m[400:] = False   # the upper earth (its mirror image lies below y = 400)
    pts = []
    for y in range(200, 396):
        xs = np.nonzero(m[y])[0]
        if xs.size > 20:
            pts += [(xs.min(), y), (xs.max(), y)]
    pts = np.array(pts, float)

    def fit(p):
        A = np.c_[2 * p[:, 0], 2 * p[:, 1], np.ones(len(p))]
        c = np.linalg.lstsq(A, (p ** 2).sum(1), rcond=None)[0]
        return c[0], c[1], float(np.sqrt(c[2] + c[0] ** 2 + c[1] ** 2))
    cx, cy, r = fit(pts)
    for _ in range(5):   # the glass ball (left) and a yellow sphere (top right) touch the silhouette: drop the rows they widen
        d = np.abs(np.hypot(pts[:, 0] - cx, pts[:, 1] - cy) - r)
        cx, cy, r = fit(pts[d < max(2.0, np.percentile(d, 70))])
    return float(cx), float(cy), float(r)


if __name__ == "__main__":
    im = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float64)
    assert im.shape == (600, 800, 3)
    cx, cy, r = find_disc(im)
    rgb = np.full((GRID, GRID, 3), np.nan)
    for j in range(GRID):          # view-space y, top row first: y = +1 - ...
        for i in range(GRID):
            vx = (i + 0.5) / GRID * 2.0 - 1.0
            vy = 1.0 - (j + 0.5) / GRID * 2.0
            if vx * vx + vy * vy > 0.93 ** 2:
                continue
            # mean over the image pixels of this cell (a cell is ~2 x 2 pixels)
            x0, x1 = cx + (vx - 1.0 / GRID) * r, cx + (vx + 1.0 / GRID) * r
            y0, y1 = cy - (vy + 1.0 / GRID) * r, cy - (vy - 1.0 / GRID) * r
            px = im[int(round(y0)):int(round(y1)) + 1, int(round(x0)):int(round(x1)) + 1].reshape(-1, 3)
            rgb[j, i] = px.mean(axis=0)
    out = os.path.join(HERE, "out_png_earth.npz")
    np.savez_compressed(out, centre_radius=np.array([cx, cy, r]), rgb=rgb.astype(np.float32))
    print(f"earth disc of {SRC}: centre ({cx:.1f}, {cy:.1f}), radius {r:.1f} px -> {out} ({os.path.getsize(out)} bytes)")
