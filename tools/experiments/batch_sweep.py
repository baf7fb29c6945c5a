#!/usr/bin/env python3
"""Development tool (GPU box, -DRT_DEV_KNOBS build for RT_BATCH_SHARE): kernel time against the number of tiles a workgroup
takes from the frame's queue per atomic ("tile_batch"), on whole frames at several samples per pixel, on the 1/2 .. 1/8
row-tile shards of the headline frame and on the other scenes.   python tools/batch_sweep.py [--lib build/ab/librt_hip_default.so]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--batches", type=int, nargs="*", default=[1, 2, 4, 8, 16])
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    if a.lib:
        pkg.hip.LIB_PATH = os.path.abspath(a.lib)
    cover = json.load(open("scenes/cfg2_cover_1200x800_spp128.json"))
    cases = []
    for spp in (8, 32, 128):
        j = dict(cover); j["samples_per_pixel"] = spp
        cases.append((f"cover spp{spp}", json.dumps(j), None))
    for world in (2, 4, 8):
        cases.append((f"cover spp128 shard 1/{world}", json.dumps(cover), (2, world - 1, world)))
    cases.append(("cfg1", open("scenes/cfg1_test_800x600_spp16.json").read(), None))
    j = json.load(open("scenes/cfg3_cover_4k_textured.json")); j["samples_per_pixel"] = 16
    cases.append(("cfg3 4K textured at spp16", json.dumps(j), None))
    for name, text, shard in cases:
        sc = pkg.host.Scene.loads(text)
        row = {"case": name}
        for b in a.batches:
            gs = pkg.hip.HipScene(sc.ptr, 0)
            gs.set_option("tile_batch", b)
            fb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
            tiles = pkg.abi.RtRowTiles(*shard) if shard else None
            ks = []
            for _ in range(a.reps):
                gs.render(fb.data_ptr(), 0, tiles, torch.cuda.current_stream().cuda_stream)
                ks.append(gs.wait()["kernel_ms"])
            ks = sorted(ks[2:])
            row[f"b{b}"] = round(ks[len(ks) // 2], 3)
            gs.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
