#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz with the CPU oracle.

The reference holds no golden images (its integration tests assert nothing, raytracer.rs:
269-284) and cannot be run here (no cargo), so these fixtures are REGRESSION vectors of the
oracle itself: they freeze the oracle's output so that a later edit of oracle/ or of the
RNG addressing cannot silently move the target the GPU path is compared against.  The
functions the oracle is built from are pinned separately by the reference's known-answer
tests (tests/test_oracle_kat.py).  Everything in the oracle is IEEE arithmetic (atan2 included: the shared
double-double routine, not libm), so the files are bit-reproducible on any machine."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

CASES = {  # name: (scene, width, height, spp, max_depth, seed)
    "cover_96x64_spp4": ("scenes/cfg2_cover_1200x800_spp128.json", 96, 64, 4, 50, 0),
    "cover_60x40_spp2_seed7": ("scenes/cfg2_cover_1200x800_spp128.json", 60, 40, 2, 50, 7),
    "test_80x60_spp4": ("scenes/cfg1_test_800x600_spp16.json", 80, 60, 4, 8, 0),
    "test_40x30_spp8_depth50": ("scenes/cfg1_test_800x600_spp16.json", 40, 30, 8, 50, 3),
    "cover_tex_64x36_spp4": ("scenes/cfg3_cover_4k_textured.json", 64, 36, 4, 50, 0),
}


def main():
    os.chdir(ROOT)
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    for name, (path, w, h, spp, depth, seed) in CASES.items():
        sc = pkg.host.Scene.load(path)
        sc.c.width, sc.c.height, sc.c.samples_per_pixel, sc.c.max_depth, sc.c.seed = w, h, spp, depth, seed
        rgb, lin, st = oracle.render(pkg.abi, sc.ptr)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), rgb8=rgb, linear=lin,
                            segments=np.uint64(st["segments"]), samples=np.uint64(st["samples"]),
                            segments_discarded=np.uint64(st["segments_discarded"]))
        print(name, rgb.shape, "segments", st["segments"], "discarded", st["segments_discarded"], "mean", lin.mean(axis=(0, 1)))


if __name__ == "__main__":
    main()
