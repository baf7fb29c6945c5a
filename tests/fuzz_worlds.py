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


def adversarial_scene(host, seed):
    """A few hundred ordinary spheres with records no JSON file can carry patched in through the C structs: NaN / +-inf /
    1e300 / denormal / zero / negative radii, NaN / inf / 1e308 centres, three coincident spheres, one light."""
    import json
    import numpy as np
    rng = np.random.default_rng(seed)
    n = 300
    objs = []
    for i in range(n):
        c = rng.uniform(-4, 4, 3)
        c[1] = rng.uniform(0, 0.6)
        kind = rng.integers(0, 10)
        mat = ({"Lambertian": {"albedo": [float(x) for x in rng.uniform(0.1, 0.9, 3)]}} if kind < 6 else
               {"Metal": {"albedo": [0.8, 0.8, 0.8], "fuzz": float(rng.uniform(0, 0.3))}} if kind < 8 else {"Glass": {"index_of_refraction": 1.5}})
        if i == 7:
            mat = {"Light": {}}
        objs.append({"center": {"x": float(c[0]), "y": float(c[1]), "z": float(c[2])}, "radius": float(rng.uniform(0.05, 0.3)), "material": mat})
    cfg = {"width": 64, "height": 40, "samples_per_pixel": 2, "max_depth": 6, "sky": {"texture": ""},
           "camera": {"look_from": {"x": 9.0, "y": 2.0, "z": 3.0}, "look_at": {"x": 0.0, "y": 0.3, "z": 0.0}, "vup": {"x": 0.0, "y": 1.0, "z": 0.0},
                      "vfov": 35.0, "aspect": 1.6}, "objects": objs}
    sc = host.Scene.loads(json.dumps(cfg))
    sp = sc.c.spheres
    odd = [("radius", float("nan")), ("radius", float("inf")), ("radius", 0.0), ("radius", 1e-310), ("radius", 1e300), ("radius", -0.2),
           ("radius", 40.0), ("cx", float("nan")), ("cy", float("-inf")), ("cz", 1e308), ("cx", float("inf"))]
    picks = rng.choice(np.arange(8, n), size=2 * len(odd), replace=False)
    for k, i in enumerate(picks):
        what, v = odd[k % len(odd)]
        if what == "radius":
            sp[i].radius = v
        else:
            sp[i].center["xyz".index(what[1])] = v
    base = min(int(picks[0]), n - 4)   # (the spheres array is a raw C pointer: stay inside it)
    for j in range(3):   # three coincident spheres (the lowest index wins ties, raytracer.rs:52-57)
        sp[base + 1 + j].center[0], sp[base + 1 + j].center[2] = 0.5, 0.5
        sp[base + 1 + j].center[1], sp[base + 1 + j].radius = 0.3, 0.3
    return sc


def big_flat_world_json(n, rng, width=12, height=8, spp=2, depth=6, half=60.0):
    """n small spheres scattered over a ground sphere (the cover scene's layer, wider: a square of side 2 * half — the cover
    scene has one sphere per unit of area): the world of the > 65 535-sphere tests"""
    import json
    objs = [{"center": {"x": 0.0, "y": -1000.0, "z": 0.0}, "radius": 1000.0, "material": {"Lambertian": {"albedo": [0.5, 0.5, 0.5]}}}]
    xs, zs = rng.uniform(-half, half, n), rng.uniform(-half, half, n)
    for i in range(n):
        m = {"Lambertian": {"albedo": [0.3, 0.6, 0.2]}} if i % 3 else ({"Metal": {"albedo": [0.8, 0.8, 0.8], "fuzz": 0.1}} if i % 2 else {"Glass": {"index_of_refraction": 1.5}})
        objs.append({"center": {"x": float(xs[i]), "y": 0.2, "z": float(zs[i])}, "radius": 0.2, "material": m})
    return json.dumps({"width": width, "height": height, "samples_per_pixel": spp, "max_depth": depth, "sky": {"texture": ""},
                       "camera": {"look_from": {"x": 13.0, "y": 2.0, "z": 3.0}, "look_at": {"x": 0.0, "y": 0.0, "z": 0.0}, "vup": {"x": 0.0, "y": 1.0, "z": 0.0},
                                  "vfov": 20.0, "aspect": 1.5}, "objects": objs})


def crowded_cell_world_json(n_crowd=4300, n_other=300, seed=9, width=24, height=16, spp=2, depth=6):
    """More spheres in ONE cell than the packed cell word can count (4 095): n_crowd nearly coincident small spheres, n_other
    ordinary ones around them.  build_grid gives such a scene the wide tables (until round 5: no grid at all)."""
    rng = np.random.default_rng(seed)
    objs = [{"center": {"x": 0.0, "y": -1000.0, "z": 0.0}, "radius": 1000.0, "material": {"Lambertian": {"albedo": [0.5, 0.5, 0.5]}}}]
    for i in range(n_crowd):
        c = rng.normal(0.0, 0.01, 3)
        objs.append({"center": {"x": float(c[0]), "y": 0.5 + float(c[1]), "z": float(c[2])}, "radius": 0.3 + 1e-4 * (i % 50),
                     "material": {"Lambertian": {"albedo": [0.2 + 0.6 * (i % 3 == 0), 0.3, 0.7]}} if i % 5 else {"Glass": {"index_of_refraction": 1.5}}})
    xs, zs = rng.uniform(-8, 8, n_other), rng.uniform(-8, 8, n_other)
    for i in range(n_other):
        objs.append({"center": {"x": float(xs[i]), "y": 0.2, "z": float(zs[i])}, "radius": 0.2,
                     "material": {"Metal": {"albedo": [0.8, 0.7, 0.6], "fuzz": 0.2}} if i % 2 else {"Lambertian": {"albedo": [0.6, 0.3, 0.2]}}})
    return json.dumps({"width": width, "height": height, "samples_per_pixel": spp, "max_depth": depth, "sky": {"texture": ""},
                       "camera": {"look_from": {"x": 6.0, "y": 1.5, "z": 2.0}, "look_at": {"x": 0.0, "y": 0.4, "z": 0.0}, "vup": {"x": 0.0, "y": 1.0, "z": 0.0},
                                  "vfov": 30.0, "aspect": 1.5}, "objects": objs})
