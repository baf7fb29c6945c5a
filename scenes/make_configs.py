#!/usr/bin/env python3
"""Derive the BASELINE.json benchmark configs from the reference's committed scene data.

Run once in the build container (needs /root/reference); the outputs are committed because
/root/reference does not exist on the GPU box.  The reference CLI has no flags — every
parameter lives in the JSON — so each config is a scene file of the reference's own schema:

  cfg1_test_800x600_spp16.json     data/test_scene.json with spp 16, depth 8            (BASELINE configs[0])
  cfg2_cover_1200x800_spp128.json  data/cover_scene.json at 1200x800, spp 128, aspect 1.5 (configs[1], headline)
  cfg3_cover_4k_textured.json      cover world at 3840x2160 spp 1024, the three r=1 spheres
                                   re-materialled to earth/moon textures, beach sky       (configs[2])
  cfg4_cover_4k_textured_spp512.json  same at spp 512 (the 8-GPU config)                  (configs[3])

Texture paths are rewritten to scenes/data/*.jpg (resolved relative to the cwd = repo root,
exactly like the reference resolves data/*.jpg relative to its cwd).  configs[4] (procedural
10 000-sphere world) is generated on the fly by scenes/procedural.py.
"""
import json
import os
import shutil

REF = "/root/reference/raytracer/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def dump(obj, name):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")


def main():
    for jpg in ("earth.jpg", "moon.jpg", "beach.jpg"):
        shutil.copyfile(os.path.join(REF, jpg), os.path.join(HERE, "data", jpg))
        os.chmod(os.path.join(HERE, "data", jpg), 0o644)

    test = json.load(open(os.path.join(REF, "test_scene.json")))
    test["samples_per_pixel"], test["max_depth"] = 16, 8
    test["sky"]["texture"] = "scenes/" + test["sky"]["texture"]
    for o in test["objects"]:
        tex = o["material"].get("Texture")
        if tex:
            tex["pixels"] = "scenes/" + tex["pixels"]
    dump(test, "cfg1_test_800x600_spp16.json")

    cover = json.load(open(os.path.join(REF, "cover_scene.json")))
    c2 = json.loads(json.dumps(cover))
    c2.update(width=1200, height=800, samples_per_pixel=128, max_depth=50)
    c2["camera"]["aspect"] = 1.5
    dump(c2, "cfg2_cover_1200x800_spp128.json")

    c3 = json.loads(json.dumps(cover))
    c3.update(width=3840, height=2160, samples_per_pixel=1024, max_depth=50)
    c3["camera"]["aspect"] = 16.0 / 9.0
    c3["sky"] = {"texture": "scenes/data/beach.jpg"}
    big = [i for i, o in enumerate(c3["objects"]) if o["radius"] == 1.0]
    assert len(big) == 3, big
    for i, jpg in zip(big, ("earth.jpg", "moon.jpg", "earth.jpg")):
        c3["objects"][i]["material"] = {"Texture": {"albedo": [1.0, 1.0, 1.0], "pixels": "scenes/data/" + jpg,
                                                    "width": 2048, "height": 1024, "h_offset": 0.75}}
    dump(c3, "cfg3_cover_4k_textured.json")
    c4 = json.loads(json.dumps(c3))
    c4["samples_per_pixel"] = 512
    dump(c4, "cfg4_cover_4k_textured_spp512.json")


if __name__ == "__main__":
    main()
