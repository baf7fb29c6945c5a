#!/usr/bin/env python3
"""Lit cover frame (cover + one light, 1200x800, spp 32) against the sizes of the two light pools (rt_core.h): kernel time and
repeated segments per forced (frames, bases) pair, then the automatic sizing.  GPU box: python tools/pool_sweep.py [--reps N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
from lit_bench import cover  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    sc = pkg.host.Scene.loads(cover(32, light=True))
    fb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
    cases = [(0, 0)] + [(f, b) for f in (32, 64, 96, 128, 160) for b in (64, 128, 192, 256, 352)] + [(0, 0)]
    for f, b in cases:
        gs = pkg.hip.HipScene(sc.ptr, 0)
        if f:
            gs.set_option("light_pool", f)
        if b:
            gs.set_option("light_base_pool", b)
        ks = []
        for _ in range(a.reps):
            gs.render(fb.data_ptr(), 0, None, torch.cuda.current_stream().cuda_stream)
            st = gs.wait()
            ks.append(st["kernel_ms"])
        print(json.dumps({"frames_forced": f, "bases_forced": b, "frames": gs.query("light_pool_slots"), "bases": gs.query("light_base_slots"),
                          "lds_tables": gs.query("lds_tables"), "lds_bytes": gs.query("lds_bytes"), "kernel_ms": round(min(ks[2:]), 3),
                          "segments_repeated": st["segments_repeated"], "segments": st["segments"]}), flush=True)
        gs.close()


if __name__ == "__main__":
    main()
