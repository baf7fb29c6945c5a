"""configs[4]: procedural random world modelled on the reference's `_make_cover_world`
(config.rs:149-226), widened from the 22x22 lattice to `half`x`half` cells per quadrant
(half=50 -> ~10 000 spheres), same 80/15/5 % material mix, same exclusion test.  The
reference draws from an unseeded thread_rng; here the draws come from numpy's Philox
bit generator with a fixed seed so the scene is reproducible."""
import json

import numpy as np


def make_world(half=50, seed=0):
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
            c = {"x": cx, "y": 0.2, "z": cz}
            if choose_mat < 0.8:
                m = {"Lambertian": {"albedo": [f32() * f32(), f32() * f32(), f32() * f32()]}}
            elif choose_mat < 0.95:
                m = {"Metal": {"albedo": [0.5 * (1 + f32()), 0.5 * (1 + f32()), 0.5 * (1 + f32())], "fuzz": 0.5 * rng.random()}}
            else:
                m = {"Glass": {"index_of_refraction": 1.5}}
            objs.append({"center": c, "radius": 0.2, "material": m})
    objs.append({"center": {"x": 0.0, "y": 1.0, "z": 0.0}, "radius": 1.0, "material": {"Glass": {"index_of_refraction": 1.5}}})
    objs.append({"center": {"x": -4.0, "y": 1.0, "z": 0.0}, "radius": 1.0, "material": {"Lambertian": {"albedo": [0.4, 0.2, 0.1]}}})
    objs.append({"center": {"x": 4.0, "y": 1.0, "z": 0.0}, "radius": 1.0, "material": {"Metal": {"albedo": [0.7, 0.6, 0.5], "fuzz": 0.0}}})
    return objs


def make_config(width=3840, height=2160, spp=2048, max_depth=50, half=50, seed=0):
    return {"width": width, "height": height, "samples_per_pixel": spp, "max_depth": max_depth,
            "sky": {"texture": ""},
            "camera": {"look_from": {"x": 13.0, "y": 2.0, "z": 3.0}, "look_at": {"x": 0.0, "y": 0.0, "z": 0.0},
                       "vup": {"x": 0.0, "y": 1.0, "z": 0.0}, "vfov": 20.0, "aspect": width / height},
            "objects": make_world(half, seed)}


def make_json(**kw):
    return json.dumps(make_config(**kw), separators=(",", ":"))


if __name__ == "__main__":
    import sys
    sys.stdout.write(make_json())
