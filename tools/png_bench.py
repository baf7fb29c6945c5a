#!/usr/bin/env python3
"""Host pipeline timing: PNG encode of RENDERED frames (the headline frame and the reference's test scene,
rendered here through the C ABI; a synthetic frame without a GPU) by deflate strategy and thread cap, JPEG decode of the textures.

    python tools/png_bench.py            # GPU box: real frames
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import __graft_entry__ as graft  # noqa: E402


def frames(pkg):
    import numpy as np
    try:
        import torch
        gpu = torch.cuda.is_available()
    except Exception:
        gpu = False
    if gpu:
        for name, path in (("cfg2_1200x800_spp128", "scenes/cfg2_cover_1200x800_spp128.json"), ("cfg1_800x600_spp16", "scenes/cfg1_test_800x600_spp16.json"),
                           ("cfg4_3840x2160_spp512", "scenes/cfg4_cover_4k_textured_spp512.json")):
            sc = pkg.host.Scene.load(path)
            img, _ = pkg.hip.render_rgb8(sc.ptr)
            yield name, img
    else:
        rng = np.random.default_rng(0)
        for w, h in ((1200, 800), (3840, 2160)):
            y, x = np.mgrid[0:h, 0:w]
            img = np.stack([x * 255 // w, y * 255 // h, (x + y) * 255 // (w + h)], -1).astype(np.int16) + rng.integers(-6, 7, (h, w, 3))
            yield f"synthetic_{w}x{h}", np.clip(img, 0, 255).astype(np.uint8)


def main():
    pkg = graft.load_package()
    host = pkg.host
    out = {"host_threads": os.cpu_count(), "png": []}
    for name, img in frames(pkg):
        for strategy in ("rle", "huffman", "default"):
            for cap in ("1", "8", "16", "32", "64", "128"):
                if int(cap) > (os.cpu_count() or 1) and cap != "1":
                    continue
                os.environ["RT_PNG_DEFLATE"], os.environ["RT_PNG_THREADS"] = strategy, cap
                ts = []
                for _ in range(7):
                    t = time.perf_counter()
                    host.png_write("/tmp/png_bench.png", img)
                    ts.append(time.perf_counter() - t)
                out["png"].append({"frame": name, "deflate": strategy, "threads_cap": int(cap), "ms_min": round(min(ts) * 1e3, 2), "ms_median": round(sorted(ts)[3] * 1e3, 2),
                                   "file_mb": round(os.path.getsize("/tmp/png_bench.png") / 1e6, 3)})
                print(out["png"][-1], file=sys.stderr, flush=True)
    del os.environ["RT_PNG_DEFLATE"], os.environ["RT_PNG_THREADS"]
    for f in ("earth.jpg", "moon.jpg", "beach.jpg"):
        p = os.path.join(ROOT, "scenes", "data", f)
        t = time.perf_counter()
        host.jpeg_decode(p)
        out[f"jpeg_{f}_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
