#!/usr/bin/env python3
"""Development tool (GPU box): kernel time of the cover scene against samples per pixel, and at one spp against the tile
size / samples per item — how much of a short frame is per-item and per-tile overhead.
    python tools/spp_sweep.py [--spp 8 16 32 64 128] [--at 32]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, nargs="*", default=[8, 16, 32, 64, 128, 256])
    ap.add_argument("--at", type=int, nargs="*", default=[32])
    ap.add_argument("--tiles", type=int, nargs="*", default=[1, 2, 3])
    ap.add_argument("--chunks", type=int, nargs="*", default=[4, 8, 16, 32])
    ap.add_argument("--scene", default="scenes/cfg2_cover_1200x800_spp128.json")
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    base = json.load(open(a.scene))

    def run(spp, opts):
        j = dict(base); j["samples_per_pixel"] = spp
        sc = pkg.host.Scene.loads(json.dumps(j))
        gs = pkg.hip.HipScene(sc.ptr, 0)
        for k, v in opts.items():
            gs.set_option(k, v)
        fb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
        ks = []
        for _ in range(a.reps):
            gs.render(fb.data_ptr(), 0, None, torch.cuda.current_stream().cuda_stream)
            st = gs.wait()
            ks.append(st["kernel_ms"])
        k = min(ks[2:])
        print(json.dumps({"spp": spp, "opts": opts, "kernel_ms": round(k, 3), "first_ms": round(ks[0], 3), "msamples_per_s": round(st["samples"] / k / 1e3, 1),
                          "ms_per_spp": round(k / spp, 4)}), flush=True)
        gs.close()

    for spp in a.spp:
        run(spp, {})
    for at in a.at:
        for tl in a.tiles:
            for ch in a.chunks:
                if ch <= at:
                    run(at, {"tile_log2": tl, "chunk_spp": ch})


if __name__ == "__main__":
    main()
