#!/usr/bin/env python3
"""Kernel time of the LIT instantiations of the megakernel (GPU box): the reference's own test_scene (BASELINE
configs[0]), the cover scene + one light at 1200x800 spp 32 (a lit scene whose tables only fit in LDS beside a POOL of
light frames), the same with the pool forced small / the option off, and an unlit cover scene with an albedo > 1 (the
<lights=0, simple_colour=0> instantiation).  python tools/lit_bench.py [--lib other.so] [--reps N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def cover(spp, light=False, hot_albedo=False):
    j = json.load(open(os.path.join(ROOT, "scenes", "cfg2_cover_1200x800_spp128.json")))
    j["samples_per_pixel"] = spp
    if light:
        j["objects"].append({"center": {"x": 0.0, "y": 30.0, "z": 10.0}, "radius": 8.0, "material": {"Light": {}}})
    if hot_albedo:   # legal JSON the reference accepts: an albedo above 1 (the clamp of raytracer.rs:118-122 then binds)
        j["objects"][5]["material"] = {"Lambertian": {"albedo": [1.5, 0.9, 0.2]}}
    return json.dumps(j)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    if a.lib:
        pkg.hip.LIB_PATH = os.path.abspath(a.lib)
    cases = [("cfg1 test_scene 800x600 spp16 depth8 (1 light, textures)", pkg.host.Scene.load("scenes/cfg1_test_800x600_spp16.json"), {}),
             ("cover 1200x800 spp32, unlit (simple colour)", pkg.host.Scene.loads(cover(32)), {}),
             ("cover 1200x800 spp32, unlit, one albedo > 1 (<lights=0, simple=0>)", pkg.host.Scene.loads(cover(32, hot_albedo=True)), {}),
             ("cover + 1 light 1200x800 spp32 (light-frame pool, tables in LDS)", pkg.host.Scene.loads(cover(32, light=True)), {}),
             ("cover + 1 light, pool forced to 32 records", pkg.host.Scene.loads(cover(32, light=True)), {"light_pool": 32}),
             ("cover + 1 light, pool forced to 64 records", pkg.host.Scene.loads(cover(32, light=True)), {"light_pool": 64})]
    for name, sc, opts in cases:
        gs = pkg.hip.HipScene(sc.ptr, 0)
        try:
            for k, v in opts.items():
                gs.set_option(k, v)
        except Exception as e:   # (an older library without the option)
            print(json.dumps({"case": name, "skipped": str(e)}))
            gs.close()
            continue
        fb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
        ks = []
        for _ in range(a.reps):
            gs.render(fb.data_ptr(), 0, None, torch.cuda.current_stream().cuda_stream)
            st = gs.wait()
            ks.append(st["kernel_ms"])
        k = min(ks[2:]) if len(ks) > 2 else min(ks)
        print(json.dumps({"case": name, "kernel_ms": round(k, 3), "msamples_per_s": round(st["samples"] / k / 1e3, 1),
                          "segments_per_sample": round(st["segments"] / st["samples"], 3), "lib": os.path.basename(pkg.hip.LIB_PATH)}), flush=True)
        gs.close()


if __name__ == "__main__":
    main()
