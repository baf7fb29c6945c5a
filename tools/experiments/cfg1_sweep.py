#!/usr/bin/env python3
"""One-process sweep of the work-distribution options on a millisecond frame (default: the reference's test_scene, cfg1):
tile size, samples per item, tiles per queue atomic, tile shape.  Median kernel time of 12 frames after 4 warm-up frames
(the queue order is re-learned after a geometry change).  GPU box:  python tools/cfg1_sweep.py [scene.json]"""
import itertools
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    sc = pkg.host.Scene.load(sys.argv[1] if len(sys.argv) > 1 else "scenes/cfg1_test_800x600_spp16.json")
    gs = pkg.hip.HipScene(sc.ptr, 0)
    rgb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream

    def run(opts):
        for k, v in opts.items():
            gs.set_option(k, v)
        ks = []
        for i in range(16):
            gs.render(rgb.data_ptr(), 0, None, stream)
            ks.append(gs.wait()["kernel_ms"])
        return statistics.median(ks[4:]), min(ks[4:])

    base = {"tile_log2": -1, "chunk_spp": 0, "tile_batch": 0, "tile_shape": 0}
    print(json.dumps({"opts": "auto", "median_min_ms": run(base)}), flush=True)
    for tl, cs in itertools.product((1, 2, 3), (2, 4, 8, 16)):
        print(json.dumps({"opts": {"tile_log2": tl, "chunk_spp": cs}, "median_min_ms": run({**base, "tile_log2": tl, "chunk_spp": cs})}), flush=True)
    for tb in (1, 2, 4, 8, 16):
        print(json.dumps({"opts": {"tile_batch": tb}, "median_min_ms": run({**base, "tile_batch": tb})}), flush=True)
    for ts in (1, 2, 3):
        print(json.dumps({"opts": {"tile_shape": ts}, "median_min_ms": run({**base, "tile_shape": ts})}), flush=True)
    for ta in (0, 2):
        print(json.dumps({"opts": {"tile_affinity": ta}, "median_min_ms": run({**base, "tile_affinity": ta})}), flush=True)
    print(json.dumps({"opts": "auto (again)", "median_min_ms": run({**base, "tile_affinity": 1})}), flush=True)


if __name__ == "__main__":
    main()
