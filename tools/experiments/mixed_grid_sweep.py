#!/usr/bin/env python3
"""One-off (round 5): the two mixed-radius 10^4-sphere worlds and the flat 2 x 10^5-sphere world against the grid builder's
knobs (cells per gridded sphere, the `large` list's radius ratio and length) — through the A/B build of the library
(build/ab/librt_hip_default.so, -DRT_DEV_KNOBS reads RT_GRID_* at scene creation).  On the GPU box:
    python tools/experiments/mixed_grid_sweep.py > gpurun_out/mixed_grid_sweep.log"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as graft  # noqa: E402


def main():
    import numpy as np
    import torch
    import procedural
    import ab_bench
    from fuzz_worlds import big_flat_world_json
    os.chdir(ROOT)
    pkg = graft.load_package()
    abi = pkg.abi
    L = ab_bench.bind(os.path.join(ROOT, "build", "ab", "librt_hip_default.so"), abi)
    L.rt_abi_version.restype = C.c_uint32
    L.rt_hip_scene_query.argtypes = [C.c_void_p, C.c_char_p]
    L.rt_hip_scene_query.restype = C.c_int64
    worlds = [("loguniform", pkg.host.Scene.loads(procedural.make_json(width=960, height=540, spp=32, half=50, seed=0, radii="loguniform"))),
              ("bimodal", pkg.host.Scene.loads(procedural.make_json(width=960, height=540, spp=32, half=50, seed=0, radii="bimodal"))),
              ("cfg5 uniform", pkg.host.Scene.loads(procedural.make_json(width=960, height=540, spp=32, half=50, seed=0))),
              ("flat 2e5", pkg.host.Scene.loads(big_flat_world_json(200000, np.random.default_rng(5), width=640, height=360, spp=16, depth=50, half=224.0)))]
    stream = torch.cuda.current_stream().cuda_stream
    st = abi.RtStats()
    ref = {}
    for cps in ("1", "2", "3", "4", "6", "8", "12"):
        for ratio, nl in (("4", None), ("2", None), ("8", None)):
            os.environ["RT_GRID_CELLS_PER_SPHERE"] = cps
            os.environ["RT_GRID_LARGE_RATIO"] = ratio
            for name, sc in worlds:
                sc.c.abi_version = L.rt_abi_version()
                hs = C.c_void_p()
                assert L.rt_hip_scene_create(sc.ptr, 0, C.byref(hs)) == 0, L.rt_hip_last_error()
                rgb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
                ks = []
                for _ in range(3):
                    assert L.rt_hip_render(hs, None, rgb.data_ptr(), None, stream) == 0, L.rt_hip_last_error()
                    assert L.rt_hip_wait(hs, C.byref(st)) == 0
                    ks.append(st.kernel_ms)
                img = rgb.cpu().numpy()
                same = name not in ref or bool((img == ref[name]).all())
                ref.setdefault(name, img)
                q = {k: L.rt_hip_scene_query(hs, k.encode()) for k in ("grid_cells", "grid_items", "grid_large", "grid_wide", "lds_tables")}
                print(f"cells/sphere {cps:>2} large ratio {ratio} | {name:13s} kernel {min(ks[1:]):8.3f} ms  tests/seg {st.exact_tests / max(1, st.segments):6.2f}  steps/seg {st.grid_steps / max(1, st.segments):5.2f}  "
                      f"cells {q['grid_cells']:>8} items {q['grid_items']:>8} large {q['grid_large']} {'same image' if same else 'IMAGE DIFFERS'}", flush=True)
                L.rt_hip_scene_destroy(hs)
                del rgb


if __name__ == "__main__":
    main()
