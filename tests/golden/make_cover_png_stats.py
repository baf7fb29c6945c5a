#!/usr/bin/env python3
"""Coarse statistics of the reference's own rendered image, raytracer/output/cover.png (800x600) — the only image-level
link to the real renderer that its unseeded RNG leaves (cover.png is ANOTHER random instance of the cover
world than the committed cover_scene.json — same camera, sky, ground and three big spheres, different small spheres —
so it is a statistical reference, not a pixel golden).

    python tests/golden/make_cover_png_stats.py      # here (reads /root/reference, which the GPU box does not have)

writes tests/golden/cover_png_stats.json: per region {box, mean RGB, median RGB}.  tests/test_gpu_parity.py renders
cover_scene.json at 800x600 on the GPU and compares the same regions within the tolerances stated there."""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/raytracer/output/cover.png"
# (x0, y0, x1, y1) in the 800x600 frame; regions whose content does not depend on the random small spheres
REGIONS = {
    "sky_band": (0, 0, 800, 35),                 # gradient sky above every sphere (raytracer.rs:142-148)
    "brown_lambertian_sphere": (235, 85, 285, 175),   # the r = 1 Lambertian (0.4, 0.2, 0.1) at (-4, 1, 0)
    "metal_sphere_sky_reflection": (470, 85, 640, 185),  # upper half of the r = 1 Metal (0.7, 0.6, 0.5): reflects the sky
    "ground_lower_third": (0, 400, 800, 600),    # ground (0.5 grey) with random small spheres: compare the MEDIAN
    "whole_image": (0, 0, 800, 600),
}


def region_stats(img, box):
    x0, y0, x1, y1 = box
    r = img[y0:y1, x0:x1].reshape(-1, 3).astype(np.float64)
    return {"box": list(box), "mean": r.mean(axis=0).round(3).tolist(), "median": np.median(r, axis=0).round(3).tolist()}


def stats_of(img):
    assert img.shape == (600, 800, 3), img.shape
    return {k: region_stats(img, b) for k, b in REGIONS.items()}


if __name__ == "__main__":
    im = np.asarray(Image.open(SRC).convert("RGB"))
    out = {"source": "raytracer/output/cover.png of the reference (800x600 RGB8)", "regions": stats_of(im)}
    with open(os.path.join(HERE, "cover_png_stats.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
