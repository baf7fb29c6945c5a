#!/usr/bin/env python3
"""Round 6 probe: kernel time of the first launches of a FRESH process (fixed queue order, same view): is a one-shot frame's
13.9 ms (against 13.2 for a scene's first frame in a warm process) a property of the process, or of the first launch?
    python tools/experiments/first_frames.py [n]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
sc = pkg.host.Scene.load("scenes/cfg2_cover_1200x800_spp128.json")
t0 = time.perf_counter()
g = pkg.hip.HipScene(sc.ptr, 0)
g.set_option("tile_order", 1)
ks = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ks.append(round(g.render_to_host()[1]["kernel_ms"], 3))
print(json.dumps({"kernel_ms_of_consecutive_frames_fixed_order": ks, "since_scene_create_ms": round((time.perf_counter() - t0) * 1e3, 1)}))
