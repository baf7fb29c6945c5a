#!/usr/bin/env python3
"""Within-process interleaved A/B of megakernel builds (cdna guide §5.4 rule 24).

    python tools/ab_bench.py build          # here (no GPU): compile the variant .so files into build/ab/
    python tools/ab_bench.py run [--rounds R] [--scene S]   # on the GPU box

Every variant must render the identical image (checked against the first one); prints one
JSON line per variant with the median/min kernel time of the headline frame."""
import ctypes as C
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

AB = os.path.join(ROOT, "build", "ab")
SRC = os.path.join(ROOT, "rust-raytracer_amd", "csrc", "hip", "rt_hip_api.hip")
BASE = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DRT_DEV_KNOBS"]  # A/B builds read the RT_GRID_* / RT_AFF_RUN_LOG2 knobs; the product does not
# name -> (extra compile flags, runtime options, environment at scene creation)
W4 = ["-DRT_WAVES_PER_EU=4"]

VARIANTS = {
    # (slow reference arms first: whatever is measured right after a 100+ ms kernel reads ~4 % high)
    "brute_force_on_gpu": (W4, {"variant": 1}, {}),
    "warmup_default": (W4, {}, {}),
    "default": (W4, {}, {}),
}
# extra arms for one session: tools/ab_arms.json {"name": [[flags], {options}, {env}, "optional source root"]} — the source
# root is another checkout of this repository (e.g. `git worktree add /tmp/prev <commit>`): the arm is built from ITS kernel
ARMS_FILE = os.path.join(ROOT, "tools", "ab_arms.json")
SRC_OF = {}
if os.path.exists(ARMS_FILE):
    for k_, v_ in json.load(open(ARMS_FILE)).items():
        VARIANTS[k_] = (list(v_[0]), dict(v_[1]), dict(v_[2]))
        if len(v_) > 3 and v_[3]:
            SRC_OF[k_] = os.path.join(v_[3], "rust-raytracer_amd", "csrc", "hip", "rt_hip_api.hip")


def build():
    os.makedirs(AB, exist_ok=True)
    built = {}
    for name, (flags, _, _) in VARIANTS.items():
        out = os.path.join(AB, f"librt_hip_{name}.so")
        key = tuple(flags) + (SRC_OF.get(name, SRC),)
        if key in built:  # same binary, different runtime options
            if os.path.lexists(out):
                os.remove(out)
            os.link(built[key], out)
            continue
        built[key] = out
        cmd = BASE + flags + [SRC_OF.get(name, SRC), "-o", out]
        print("+", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)


def bind(path, abi):
    L = C.CDLL(path)
    L.rt_hip_scene_create.argtypes = [C.POINTER(abi.RtScene), C.c_int, C.POINTER(C.c_void_p)]
    L.rt_hip_scene_destroy.argtypes = [C.c_void_p]
    L.rt_hip_scene_destroy.restype = None
    L.rt_hip_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_hip_wait.argtypes = [C.c_void_p, C.POINTER(abi.RtStats)]
    L.rt_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.rt_hip_last_error.restype = C.c_char_p
    return L


def run(rounds, scene_path, only):
    import numpy as np
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    abi = pkg.abi
    sc = pkg.host.Scene.load(scene_path)
    h, w = sc.c.height, sc.c.width
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    libs, ref = {}, {}
    for name, (_, opts, env) in VARIANTS.items():
        if only and name not in only:
            continue
        path = os.path.join(AB, f"librt_hip_{name}.so")
        if not os.path.exists(path):
            continue
        L = bind(path, abi)
        hs = C.c_void_p()
        os.environ.update(env)
        L.rt_abi_version.restype = C.c_uint32
        sc.c.abi_version = L.rt_abi_version()   # (an arm built from an older checkout: RtScene's layout has not changed since v3)
        assert L.rt_hip_scene_create(sc.ptr, 0, C.byref(hs)) == 0, L.rt_hip_last_error()
        for k in env:
            del os.environ[k]
        for k, v in opts.items():
            assert L.rt_hip_set_option(hs, k.encode(), v) == 0
        libs[name] = (L, hs, [])
    st = abi.RtStats()
    stats = {}
    for r in range(rounds + 1):  # round 0 = warm-up + image check
        for name, (L, hs, times) in libs.items():
            assert L.rt_hip_render(hs, None, rgb.data_ptr(), None, stream) == 0, L.rt_hip_last_error()
            assert L.rt_hip_wait(hs, C.byref(st)) == 0
            if r == 0:
                img = rgb.cpu().numpy()
                ref.setdefault(0, img)
                if not os.environ.get("AB_ALLOW_DIFFERENT"):  # (timing-only arms that render a WRONG image on purpose set it)
                    assert np.array_equal(img, ref[0]), f"{name}: image differs from the first variant"
            else:
                times.append(st.kernel_ms)
            stats[name] = (st.exact_tests, st.grid_steps, st.segments)
    samples = w * h * sc.c.samples_per_pixel
    for name, (L, hs, times) in libs.items():
        med = statistics.median(times)
        print(json.dumps({"variant": name, "kernel_ms_median": round(med, 3), "kernel_ms_min": round(min(times), 3),
                          "msamples_per_s": round(samples / med / 1e3, 1), "rounds": rounds,
                          "exact_per_segment": round(stats[name][0] / max(1, stats[name][2]), 3),
                          "steps_per_segment": round(stats[name][1] / max(1, stats[name][2]), 3)}), flush=True)
        L.rt_hip_scene_destroy(hs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("cmd")
        ap.add_argument("--rounds", type=int, default=5)
        ap.add_argument("--scene", default="scenes/cfg2_cover_1200x800_spp128.json")
        ap.add_argument("--only", nargs="*", default=None)
        a = ap.parse_args()
        run(a.rounds, a.scene, a.only)
