#!/usr/bin/env python3
"""Host pipeline timing: PNG encode of a rendered-looking frame, JPEG decode of the textures."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import numpy as np
    host = graft.load_package().host
    rng = np.random.default_rng(0)
    out = {"host_threads": os.cpu_count()}
    for w, h in ((1200, 800), (3840, 2160)):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([x * 255 // w, y * 255 // h, (x + y) * 255 // (w + h)], -1).astype(np.int16) + rng.integers(-6, 7, (h, w, 3))
        img = np.clip(img, 0, 255).astype(np.uint8)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            host.png_write("/tmp/png_bench.png", img)
            best = min(best, time.perf_counter() - t)
        out[f"png_{w}x{h}_ms"] = round(best * 1e3, 1)
        out[f"png_{w}x{h}_mb"] = round(os.path.getsize("/tmp/png_bench.png") / 1e6, 2)
    for f in ("earth.jpg", "moon.jpg", "beach.jpg"):
        p = os.path.join(ROOT, "scenes", "data", f)
        t = time.perf_counter()
        host.jpeg_decode(p)
        out[f"jpeg_{f}_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
