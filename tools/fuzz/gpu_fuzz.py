#!/usr/bin/env python3
"""Development tool (GPU box): many adversarial / random worlds through the megakernel — grid walk (variant 0) against
the reference's brute force on the GPU (variant 1), bit for bit, and both against the CPU oracle's NaN mask and values.
    python tools/fuzz/gpu_fuzz.py [first_seed] [n_seeds]
FUZZ_WIDE=1: every world through the WIDE cell tables (32-bit item lists — what scenes of more than 65 535 spheres use) and the
kernel's wide instantiations: librt_hip_probe.so with RT_GRID_WIDE=1 (the product library reads nothing from the environment)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    from fuzz_worlds import adversarial_scene, fuzz_world_json
    import oracle as orc
    os.chdir(ROOT)
    pkg = graft.load_package()
    abi, host, hip = pkg.abi, pkg.host, pkg.hip
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad = 0
    wide = os.environ.get("FUZZ_WIDE") == "1"
    library = None
    if wide:
        os.environ["RT_GRID_WIDE"] = "1"
        library = hip.probe_lib()
    kinds = ["adversarial"] + [f"fuzz{k}" for k in range(6)]
    for seed in range(first, first + n):
        kind = kinds[seed % len(kinds)]
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"seed {seed} {kind}", flush=True)
        if kind == "adversarial":
            sc = adversarial_scene(host, seed)
        else:
            sc = host.Scene.loads(fuzz_world_json(np.random.default_rng(seed), int(kind[4:])))
            sc.c.seed = seed
        h, w = sc.c.height, sc.c.width
        imgs = []
        for variant in (0, 1):
            gs = hip.HipScene(sc.ptr, 0, library=library)
            if wide and kind != "adversarial":
                assert gs.query("grid_wide") == 1
            gs.set_option("variant", variant)
            gs.set_option("tile_log2", [-1, 0, 1, 2, 3][seed % 5])
            gs.set_option("tile_batch", [0, 1, 2, 7, 64, 3, 0][seed % 7])       # tiles per queue atomic (the workgroup's stash)
            gs.set_option("light_pool", [0, 32, 64, 0, 96, 0][seed % 6])   # lit worlds: small frame pools force repeated segments and HBM overflows,
            gs.set_option("light_base_pool", [0, 0, 32, 64, 0][seed % 5])   # small base pools repeats,
            gs.set_option("light_nest_pool", [1, 1, 0][seed % 3])           # 0: every nested light activation through the HBM overflow
            rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
            lin = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
            gs.render(rgb.data_ptr(), lin.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            st = gs.wait()
            imgs.append((rgb.cpu().numpy(), lin.cpu().numpy(), st))
        (r0, l0, s0), (r1, l1, s1) = imgs
        same = np.array_equal(r0, r1) and np.array_equal(np.isnan(l0), np.isnan(l1)) and np.array_equal(np.nan_to_num(l0), np.nan_to_num(l1)) and s0["segments"] == s1["segments"]
        o_rgb, o_lin, o_st = orc.render(abi, sc.ptr)
        nan = np.isnan(o_lin)
        vs_oracle = np.array_equal(np.isnan(l0), nan) and float(np.abs(np.where(nan, 0, l0) - np.where(nan, 0, o_lin)).max()) <= 4e-6
        if not (same and vs_oracle):
            bad += 1
            print(f"seed {seed} kind {kind}: grid==brute {same}, vs oracle {vs_oracle}", flush=True)
    print(f"gpu_fuzz{' (wide tables)' if wide else ''}: seeds {first}..{first + n - 1}: {bad} mismatching worlds")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
