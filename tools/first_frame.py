#!/usr/bin/env python3
"""Kernel time of a scene's FIRST frame (a fresh RtHipScene each time: what a one-shot rt_render_rgb8 gets) under the
queue orders a frame without a measured order can take, beside the steady state (third frame of one scene): whole headline
frame and its 1/8 shard.  python tools/first_frame.py [--fresh N]  (GPU box)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

ORDERS = [("bottom row first (tile_order 1)", {"tile_order": 1}),
          ("top row first (tile_order 0)", {"tile_order": 0}),
          ("projection seed (tile_order 3, order_seed 1)", {"tile_order": 3, "order_seed": 1}),
          ("probe launch (tile_order 3, order_seed 2)", {"tile_order": 3, "order_seed": 2})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fresh", type=int, default=6)
    ap.add_argument("--scene", default="scenes/cfg2_cover_1200x800_spp128.json")
    a = ap.parse_args()
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    sc = pkg.host.Scene.load(a.scene)
    stream = torch.cuda.current_stream().cuda_stream
    for what, tiles in (("whole frame", None), ("1/8 shard (rank 3, 2-row interleave)", pkg.abi.RtRowTiles(2, 3, 8))):
        rows = pkg.abi.tiles_local_rows(sc.c.height, tiles)
        fb = torch.zeros((rows, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
        ref = None
        for name, opts in ORDERS:
            ks = []
            for _ in range(a.fresh):
                gs = pkg.hip.HipScene(sc.ptr, 0)
                for k, v in opts.items():
                    gs.set_option(k, v)
                gs.render(fb.data_ptr(), 0, tiles, stream)
                ks.append(gs.wait()["kernel_ms"])
                gs.close()
            img = fb.cpu().numpy()
            ref = img if ref is None else ref
            assert (img == ref).all(), name
            print(json.dumps({"frame": what, "order": name, "first_frame_kernel_ms_min": round(min(ks[1:]), 4), "median": round(sorted(ks[1:])[len(ks[1:]) // 2], 4)}), flush=True)
        gs = pkg.hip.HipScene(sc.ptr, 0)
        ks = []
        for _ in range(6):
            gs.render(fb.data_ptr(), 0, tiles, stream)
            ks.append(gs.wait()["kernel_ms"])
        gs.close()
        print(json.dumps({"frame": what, "order": "steady state (measured order, default options): frames 1..6 of one scene", "kernel_ms": [round(k, 4) for k in ks]}), flush=True)


if __name__ == "__main__":
    main()
