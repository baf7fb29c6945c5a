"""configs[4]: procedural random world modelled on the reference's `_make_cover_world`
(config.rs:149-226), widened from the 22x22 lattice to `half`x`half` cells per quadrant
(half=50 -> ~10 000 spheres), same 80/15/5 % material mix, same exclusion test.  The
reference draws from an unseeded thread_rng; here the draws come from numpy's Philox
bit generator with a fixed seed so the scene is reproducible.

`radii` (round 5, SURVEY §8 f2 "one speed number outside BASELINE's sphere distribution"): "uniform" = every small sphere
r = 0.2 (configs[4]); "loguniform" = radii log-uniform in [0.05, 5]; "bimodal" = 95 % r = 0.05, 5 % r = 3.0 — both on the
same jittered lattice stretched by 4 (so that the big spheres leave sky between them), every sphere resting on the ground
(centre y = r), overlaps allowed: worlds a single-level uniform grid is NOT made for (a cell sized for the small spheres is
crossed by every big one)."""
import json

import numpy as np


def make_world(half=50, seed=0, radii="uniform"):
    rng = np.random.Generator(np.random.Philox(seed))
    objs = [{"center": {"x": 0.0, "y": -1000.0, "z": 0.0}, "radius": 1000.0,
             "material": {"Lambertian": {"albedo": [0.5, 0.5, 0.5]}}}]
    f32 = lambda: float(np.float32(rng.random()))
    for a in range(-half, half):
        for b in range(-half, half):
            choose_mat = rng.random()
            cx, cz = a + 0.9 * rng.random(), b + 0.9 * rng.random()
            if ((cx - 4.0) ** 2 + (0.2 - 0.2) ** 2 + cz ** 2) ** 0.5 < 0.9:
                continue
            if radii != "uniform":
                cx, cz = 4.0 * cx, 4.0 * cz
            if radii == "loguniform":
                r = float(np.exp(rng.uniform(np.log(0.05), np.log(5.0))))
            elif radii == "bimodal":
                r = 3.0 if rng.random() < 0.05 else 0.05
            else:
                r = 0.2
            if radii != "uniform" and ((cx - 13.0) ** 2 + (r - 2.0) ** 2 + (cz - 3.0) ** 2) ** 0.5 < r + 0.5:
                continue   # (a big sphere around make_config's camera: every path would bounce inside it to max_depth)
            c = {"x": cx, "y": r, "z": cz}
            if choose_mat < 0.8:
                m = {"Lambertian": {"albedo": [f32() * f32(), f32() * f32(), f32() * f32()]}}
            elif choose_mat < 0.95:
                m = {"Metal": {"albedo": [0.5 * (1 + f32()), 0.5 * (1 + f32()), 0.5 * (1 + f32())], "fuzz": 0.5 * rng.random()}}
            else:
                m = {"Glass": {"index_of_refraction": 1.5}}
            objs.append({"center": c, "radius": r, "material": m})
    objs.append({"center": {"x": 0.0, "y": 1.0, "z": 0.0}, "radius": 1.0, "material": {"Glass": {"index_of_refraction": 1.5}}})
    objs.append({"center": {"x": -4.0, "y": 1.0, "z": 0.0}, "radius": 1.0, "material": {"Lambertian": {"albedo": [0.4, 0.2, 0.1]}}})
    objs.append({"center": {"x": 4.0, "y": 1.0, "z": 0.0}, "radius": 1.0, "material": {"Metal": {"albedo": [0.7, 0.6, 0.5], "fuzz": 0.0}}})
    return objs


def make_config(width=3840, height=2160, spp=2048, max_depth=50, half=50, seed=0, radii="uniform"):
    return {"width": width, "height": height, "samples_per_pixel": spp, "max_depth": max_depth,
            "sky": {"texture": ""},
            "camera": {"look_from": {"x": 13.0, "y": 2.0, "z": 3.0}, "look_at": {"x": 0.0, "y": 0.0, "z": 0.0},
                       "vup": {"x": 0.0, "y": 1.0, "z": 0.0}, "vfov": 20.0, "aspect": width / height},
            "objects": make_world(half, seed, radii)}


def make_json(**kw):
    return json.dumps(make_config(**kw), separators=(",", ":"))


if __name__ == "__main__":
    import sys
    sys.stdout.write(make_json())
