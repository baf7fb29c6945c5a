#!/usr/bin/env python3
"""Round 6 experiment: does the NEXT frame's kernel, launched on a second stream from a second resident scene, fill the tail of the
current one?  A frame ends on its deepest paths while CUs whose workgroups found the queue empty sit idle; kernels on different HIP
streams are independent, and a persistent workgroup of frame i + 1 can start on every CU frame i has left.
    python tools/experiments/overlap_tail.py [--scene S] [--frames 40] [--order 1|2] [--tiles r,G]
Prints frames/s of K frames back to back on ONE stream / alternating over TWO streams (two scenes), same images asserted."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="scenes/cfg2_cover_1200x800_spp128.json")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--order", type=int, default=1, help="tile_order: 1 = fixed (what a new view gets), 2 = learned from the previous frame of the view")
    ap.add_argument("--shard", default="", help="rank,world: one rank's interleaved 2-scanline tiles")
    a = ap.parse_args()
    import torch
    pkg = graft.load_package()
    sc = pkg.host.Scene.load(a.scene)
    tiles = None
    if a.shard:
        r, w = (int(v) for v in a.shard.split(","))
        tiles = pkg.abi.RtRowTiles(2, r, w)
    rows = pkg.abi.tiles_local_rows(sc.c.height, tiles)
    scenes = [pkg.hip.HipScene(sc.ptr, 0) for _ in range(2)]
    for g in scenes:
        g.set_option("tile_order", a.order)
    streams = [torch.cuda.Stream() for _ in range(2)]
    fbs = [torch.zeros((rows, sc.c.width, 3), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    res = {}
    for mode in ("one_stream", "two_streams", "one_stream", "two_streams"):
        n = 1 if mode == "one_stream" else 2
        for k in range(4):   # warm-up (and the learned order of --order 2)
            scenes[k % n].render(fbs[k % n].data_ptr(), 0, tiles, streams[k % n].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.frames):
            scenes[k % n].render(fbs[k % n].data_ptr(), 0, tiles, streams[k % n].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for g in scenes[:n]:
            g.wait()
        res.setdefault(mode, []).append(round(a.frames / dt, 2))
    same = bool(torch.equal(fbs[0], fbs[1]))
    print(json.dumps({"scene": os.path.basename(a.scene), "shard": a.shard, "tile_order": a.order, "frames": a.frames, "frames_per_s": res,
                      "gain": round(max(res["two_streams"]) / max(res["one_stream"]), 4), "images_equal": same}))


if __name__ == "__main__":
    main()
