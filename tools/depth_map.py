#!/usr/bin/env python3
"""Where the deep paths are: the per-tile maximum path depth the first frame of the headline scene measures (what the
steady-state queue order is sorted by), as a coarse character map and a histogram.  GPU box."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import numpy as np
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    sc = pkg.host.Scene.load("scenes/cfg2_cover_1200x800_spp128.json")
    gs = pkg.hip.HipScene(sc.ptr, 0, library=pkg.hip.probe_lib())   # (rt_hip_debug_tile_depth: include/rt_abi_test.h)
    fb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
    gs.render(fb.data_ptr(), 0, None, torch.cuda.current_stream().cuda_stream)
    gs.wait()
    d = gs.debug_tile_depth()
    gs.close()
    print("tile grid", d.shape, "histogram of max depth per tile:")
    h = np.bincount(np.minimum(d.ravel(), 50), minlength=51)
    print(json.dumps({str(k): int(v) for k, v in enumerate(h) if v}))
    ty, tx = d.shape
    by, bx = ty // 40, tx // 100
    m = d[: by * 40, : bx * 100].reshape(40, by, 100, bx).max(axis=(1, 3))
    chars = " .:-=+*#%@"
    for row in m:
        print("".join(chars[min(9, int(v) // 6)] for v in row))
    frac = [(t, float((d >= t).mean())) for t in (2, 10, 20, 30, 40, 50)]
    print("fraction of tiles with max depth >= t:", frac)


if __name__ == "__main__":
    main()
