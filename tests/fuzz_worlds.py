"""Random worlds shared by the CPU grid audit and the GPU grid-vs-brute-force test."""
import json

import numpy as np


def fuzz_world_json(rng, kind, w=72, h=48, spp=3, depth=12):
    """Random worlds that stress the grid walk's wave-level code (lock-step rounds, inline item
    pairs, last-sphere mailbox, EXIT border, `large` list, fallback): mixed materials, overlapping /
    nested / duplicated spheres, a 1e3 scale range, a camera inside glass, a world far from the origin."""
    objs = []

    def mat():
        r = rng.random()
        if r < 0.5:
            return {"Lambertian": {"albedo": [round(float(v), 3) for v in rng.random(3)]}}
        if r < 0.75:
            return {"Metal": {"albedo": [round(float(v), 3) for v in rng.uniform(0.3, 1.0, 3)], "fuzz": round(float(rng.choice([0.0, rng.random()])), 3)}}
        return {"Glass": {"index_of_refraction": round(float(rng.uniform(1.1, 2.4)), 3)}}

    def add(c, r, m=None):
        objs.append({"center": {"x": float(c[0]), "y": float(c[1]), "z": float(c[2])}, "radius": float(r), "material": m or mat()})

    off = np.zeros(3)
    look_from, look_at = np.array([6.0, 2.0, 5.0]), np.array([0.0, 0.5, 0.0])
    if kind == 0:    # dense overlapping cluster, duplicates and concentric shells (ties go to the lower index)
        for _ in range(150):
            c = rng.normal(0, 1.2, 3); r = rng.uniform(0.05, 0.7)
            add(c, r)
            if rng.random() < 0.2:
                add(c, r)                       # exact duplicate
            if rng.random() < 0.2:
                add(c, -0.8 * r, {"Glass": {"index_of_refraction": 1.5}})   # hollow shell inside (negative radius)
    elif kind == 1:  # three decades of radii on a ground ball
        add([0, -1000, 0], 1000, {"Lambertian": {"albedo": [0.5, 0.5, 0.5]}})
        for _ in range(250):
            r = 10.0 ** rng.uniform(-2.3, 0.3)
            add([rng.uniform(-6, 6), r, rng.uniform(-6, 6)], r)
    elif kind == 2:  # camera inside a big glass ball that also contains small spheres
        add([6.0, 2.0, 5.0], 3.0, {"Glass": {"index_of_refraction": 1.5}})
        for _ in range(120):
            add(rng.uniform(-5, 8, 3), rng.uniform(0.1, 0.5))
    elif kind == 3:  # one flat layer of equal spheres (grid one cell high) plus a few tall outliers
        for a in range(-9, 9):
            for b in range(-9, 9):
                add([a + 0.5 * rng.random(), 0.2, b + 0.5 * rng.random()], 0.2)
        for _ in range(4):
            add([rng.uniform(-5, 5), 3.0, rng.uniform(-5, 5)], 3.0)
    elif kind == 4:  # far from the origin: f32 cell arithmetic at large coordinates
        off = np.array([4000.0, -2500.0, 7000.0])
        for _ in range(200):
            add(off + rng.uniform(-4, 4, 3), rng.uniform(0.1, 0.6))
    else:            # sparse world with long empty walks and spheres touching each other
        for i in range(60):
            c = rng.uniform(-15, 15, 3); r = rng.uniform(0.2, 1.0)
            add(c, r)
            add(c + np.array([2 * r, 0, 0]), r)   # tangent twin
    # two lights (the reference's nested light sampling grows like n_lights^depth: raytracer.rs:92-114)
    for i in rng.choice(len(objs), 2, replace=False):
        if "Glass" not in objs[int(i)]["material"] or objs[int(i)]["radius"] > 0:
            objs[int(i)]["material"] = {"Light": {}}
    cam = {"look_from": dict(zip("xyz", (look_from + off).tolist())), "look_at": dict(zip("xyz", (look_at + off).tolist())),
           "vup": {"x": 0.0, "y": 1.0, "z": 0.0}, "vfov": 50.0, "aspect": w / h}
    return json.dumps({"width": w, "height": h, "samples_per_pixel": spp, "max_depth": depth, "sky": {"texture": ""},
                       "camera": cam, "objects": objs})
