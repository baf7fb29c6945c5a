"""Parity tests proper: the HIP megakernel, called through the C ABI (librt_hip.so), against
the CPU oracle on identical Philox seeds, against the committed golden fixtures, and — at
BASELINE.json's full size — through size-independent properties.  Needs a real MI355X."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from fuzz_worlds import fuzz_world_json
from conftest import dvec
from parity import assert_parity, pooled_atol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "scenes"))

CASES = {
    "cover_96x64_spp4": ("cover", 96, 64, 4, 50, 0),
    "cover_60x40_spp2_seed7": ("cover", 60, 40, 2, 50, 7),
    "test_80x60_spp4": ("test", 80, 60, 4, 8, 0),
    "test_40x30_spp8_depth50": ("test", 40, 30, 8, 50, 3),
    "cover_tex_64x36_spp4": ("cover4k_tex", 64, 36, 4, 50, 0),
}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def gpu_render(pkg, abi, torch_cuda):
    torch = torch_cuda

    def _render(scene, tiles=None, variant=0, want_linear=True, pool=None, chunk_spp=None, tile_log2=None):
        """variant 0: the product kernel (grid walk, tile queue, exact fixed-point pixel sums);
        variant 1: same kernel, brute force over all spheres; variant 2: the round-1 cull-scan
        kernel, where pool=0 selects one lane per pixel with sequential f32 sums (the
        reference's summation order).  chunk_spp: samples of a pixel per work item; tile_log2:
        pixel tiles of 2^k x 2^k."""
        sc = scene.c
        rows = abi.tiles_local_rows(sc.height, tiles)
        gs = pkg.hip.HipScene(scene.ptr, 0)
        if variant:
            gs.set_option("variant", variant)
        if pool is not None:
            gs.set_option("pool", pool)
        if chunk_spp is not None:
            gs.set_option("chunk_spp", chunk_spp)
        if tile_log2 is not None:
            gs.set_option("tile_log2", tile_log2)
        rgb = torch.zeros((rows, sc.width, 3), dtype=torch.uint8, device="cuda:0")
        lin = torch.zeros((rows, sc.width, 3), dtype=torch.float32, device="cuda:0") if want_linear else None
        gs.render(rgb.data_ptr(), lin.data_ptr() if want_linear else 0, tiles, torch.cuda.current_stream().cuda_stream)
        st = gs.wait()
        out = rgb.cpu().numpy(), (lin.cpu().numpy() if want_linear else None), st
        gs.close()
        return out
    return _render


def test_gpu_math_is_ieee_exact(pkg, torch_cuda):
    """bit-parity with the CPU oracle needs correctly rounded f64 sqrt/div and f32 sqrt on
    the GPU; atan2 (libm vs ocml) is allowed to differ in the last ulp (texture u only)."""
    torch = torch_cuda
    n = 1 << 20
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(n // 2), 10.0 ** rng.uniform(-30, 30, n // 2)])
    # the library's square root rescales arguments below 2^-767: both sides of that threshold, denormals, the
    # largest finite values, perfect squares and their neighbours, 0 and inf
    k = 4096
    sq = rng.integers(1, 1 << 26, k).astype(np.float64) ** 2
    x[:12 * k] = np.concatenate([2.0 ** rng.uniform(-780, -755, k), 2.0 ** rng.uniform(-1074, -1000, k), 2.0 ** rng.uniform(1000, 1023.99, k),
                                 2.0 ** rng.uniform(-1022, 1023, 4 * k), sq, np.nextafter(sq, 0), np.nextafter(sq, np.inf),
                                 np.array([0.0, 2.0 ** -767, np.nextafter(2.0 ** -767, 0), np.finfo(np.float64).max, np.inf, 5e-324] * (k // 6) + [0.0] * (k % 6)),
                                 rng.random(k) * 1e-300])
    y = np.concatenate([rng.random(n // 2) + 1e-3, 10.0 ** rng.uniform(-30, 30, n // 2)])
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    o_sqrt, o_div, o_at = torch.empty_like(dx), torch.empty_like(dx), torch.empty_like(dx)
    o_sqrtf = torch.empty(n, dtype=torch.float32, device="cuda")
    rc = pkg.hip.lib().rt_hip_math_probe(dx.data_ptr(), dy.data_ptr(), o_sqrt.data_ptr(), o_div.data_ptr(), o_sqrtf.data_ptr(),
                                         o_at.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(o_sqrt.cpu().numpy(), np.sqrt(x))
    with np.errstate(all="ignore"):
        assert np.array_equal(o_div.cpu().numpy(), x / y, equal_nan=True)
    with np.errstate(all="ignore"):
        assert np.array_equal(o_sqrtf.cpu().numpy(), np.sqrt(x.astype(np.float32)))
    fin = np.isfinite(x)
    at, want = o_at.cpu().numpy()[fin], np.arctan2(x - 0.5, y - 0.5)[fin]
    assert np.allclose(at, want, rtol=4e-16, atol=0)
    print("atan2 last-ulp mismatches:", float((at != want).mean()))


@pytest.mark.parametrize("name", sorted(CASES))
def test_matches_oracle_and_golden(name, gpu_render, oracle, hostsim, abi, load_scene):
    scene, w, h, spp, depth, seed = CASES[name]
    sc = load_scene(scene, w, h, spp, depth, seed)
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    # (1) round-1 kernel in the reference's summation order (one lane per pixel): the tight tolerance
    rgb, lin, st = gpu_render(sc, variant=2, pool=0)
    err, flips = assert_parity(rgb, lin, o_rgb, o_lin, name + " vs oracle")
    if "tex" in name or name.startswith("test_"):
        # libm-vs-ocml atan2 can move a texel: allow isolated sample-level differences vs the frozen file
        assert np.abs(lin - g["linear"]).max() <= 0.05 and (rgb != g["rgb8"]).mean() < 1e-3
    else:
        assert_parity(rgb, lin, g["rgb8"], g["linear"], name + " vs golden")
    # (2) the product kernel: same paths (grid walk = the reference's closest hit), exact fixed-point pixel sums
    p_rgb, p_lin, p_st = gpu_render(sc)
    p_err, p_flips = assert_parity(p_rgb, p_lin, o_rgb, o_lin, name + " product vs oracle", atol=pooled_atol(spp), flip_frac=5e-4)
    h_rgb, h_lin, h_st = hostsim.render(sc.ptr, None, 3 + 16)  # CPU build of the same per-lane code, fixed-point sums
    if not ("tex" in name or name.startswith("test_")):    # integer sums are order-free: bit-exact without libm in the path
        assert np.array_equal(p_lin, h_lin) and np.array_equal(p_rgb, h_rgb)
        assert p_st["exact_tests"] == h_st["exact_tests"] and p_st["grid_steps"] == h_st["grid_steps"]
    # splitting a pixel's samples over several work items (HBM accumulator + epilogue) changes no bit
    for cs, tl in ((1, 3), (3, 2), (spp, 1), (2, 0), (spp, 3)):  # chunking and pixel-tile size change no bit either
        c_rgb, c_lin, c_st = gpu_render(sc, chunk_spp=cs, tile_log2=tl)
        assert np.array_equal(c_rgb, p_rgb) and np.array_equal(c_lin, p_lin) and c_st["segments"] == p_st["segments"], (cs, tl)
    for s_ in (st, p_st):
        assert s_["samples"] == w * h * spp == o_st["samples"]
        if sc.lights():
            assert 0 < s_["segments"] <= o_st["segments"]
        else:
            assert s_["segments"] == o_st["segments"] == int(g["segments"])
        assert s_["sphere_tests"] == s_["segments"] * sc.c.n_spheres and s_["tex_oob"] == 0
    print(f"{name}: max|dlin|={p_err:.2e} (round-1 kernel, f32 sums {err:.2e}) rgb8 flips={p_flips} ({flips}) "
          f"exact/segment={p_st['exact_tests'] / max(1, p_st['segments']):.2f} steps/segment={p_st['grid_steps'] / max(1, p_st['segments']):.2f}")


@pytest.mark.parametrize("scene,w,h,spp,depth", [("cover", 64, 48, 3, 50), ("test", 48, 36, 3, 8)])
def test_grid_variant_equals_bruteforce_variant(gpu_render, load_scene, scene, w, h, spp, depth):
    """variant 1 runs the reference's exact test on every sphere (no grid), variant 2 is the
    round-1 cull-scan kernel: all three must produce the same bits."""
    sc = load_scene(scene, w, h, spp, depth)
    a_rgb, a_lin, a_st = gpu_render(sc, variant=0)
    b_rgb, b_lin, b_st = gpu_render(sc, variant=1)
    c_rgb, c_lin, c_st = gpu_render(sc, variant=2, pool=1)
    assert np.array_equal(a_rgb, b_rgb) and np.array_equal(a_lin, b_lin)
    assert np.array_equal(a_rgb, c_rgb) and np.array_equal(a_lin, c_lin)
    assert a_st["segments"] == b_st["segments"] == c_st["segments"]
    assert b_st["exact_tests"] == b_st["sphere_tests"] >= a_st["exact_tests"] and b_st["grid_steps"] == 0
    if sc.c.n_spheres > 64:
        assert a_st["grid_steps"] > 0 and a_st["exact_tests"] < 0.05 * a_st["sphere_tests"]


def test_work_distribution_stress(gpu_render, load_scene):
    """The tile-slot protocol under contention: thousands of one-pixel tiles with one-sample chunks
    (every acquire opens or re-opens a slot), big tiles with tiny chunks, ragged image edges — the
    frame and the path count never change, run after run."""
    sc = load_scene("cover", 203, 117, 5, 50)  # neither dimension a multiple of any tile size
    ref_rgb, ref_lin, ref_st = gpu_render(sc)
    for tl, cs in ((0, 1), (0, 5), (1, 1), (2, 2), (3, 1), (3, 5), (1, 3)):
        for _ in range(2):
            rgb, lin, st = gpu_render(sc, chunk_spp=cs, tile_log2=tl)
            assert np.array_equal(rgb, ref_rgb) and np.array_equal(lin, ref_lin), (tl, cs)
            assert st["segments"] == ref_st["segments"] and st["samples"] == 203 * 117 * 5


@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_tile_shards_reassemble_bit_identically(gpu_render, abi, load_scene, world):
    sc = load_scene("cover", 72, 45, 3, 50)
    full_rgb, full_lin, _ = gpu_render(sc)
    seen = 0
    for rank in range(world):
        t = abi.RtRowTiles(2 if world == 8 else 8, rank, world)
        rows = abi.tiles_global_rows(45, t)
        rgb, lin, st = gpu_render(sc, tiles=t)
        assert np.array_equal(rgb, full_rgb[rows]) and np.array_equal(lin, full_lin[rows])
        assert st["samples"] == len(rows) * 72 * 3
        seen += len(rows)
    assert seen == 45


def test_degenerate_scenes(gpu_render, oracle, abi, host):
    base = ('{"width":9,"height":5,"samples_per_pixel":3,"max_depth":%d,"sky":%s,"camera":{"look_from":{"x":0.0,"y":0.0,"z":0.0},'
            '"look_at":{"x":0.0,"y":0.0,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":90.0,"aspect":1.8},"objects":[%s]}')
    lam = '{"center":{"x":0.0,"y":0.0,"z":-1.0},"radius":0.5,"material":{"Lambertian":{"albedo":[0.8,0.3,0.3]}}}'
    light = '{"center":{"x":0.0,"y":3.0,"z":-1.0},"radius":1.0,"material":{"Light":{}}}'
    bright = '{"center":{"x":0.0,"y":-100.5,"z":-1.0},"radius":100.0,"material":{"Metal":{"albedo":[1.5,0.2,2.0],"fuzz":0.3}}}'
    for depth in (0, 1, 2, 5):
        for sky in ("null", '{"texture":""}'):
            for objs in ("", lam, lam + "," + light, light + "," + lam + "," + light + "," + bright):
                sc = host.Scene.loads(base % (depth, sky, objs))
                o_rgb, o_lin, _ = oracle.render(abi, sc.ptr)
                for variant, pool in ((0, None), (2, 0)):
                    rgb, lin, _ = gpu_render(sc, variant=variant, pool=pool)
                    assert_parity(rgb, lin, o_rgb, o_lin, f"depth {depth} sky {sky} objs {len(objs)} variant {variant}")


@pytest.mark.parametrize("w,h", [(1, 1), (1, 4), (5, 1)])
def test_one_pixel_wide_or_high_frames(gpu_render, oracle, abi, host, w, h):
    """width - 1 = 0 or height - 1 = 0: the reference divides by zero (raytracer.rs:199-200), every ray is
    NaN/inf, every sample NaN, and palette turns the NaN pixel into 0.  The product's RGB8 is the same; its
    diagnostic linear image holds 0 where the oracle's holds NaN (a NaN sample adds 0 to the fixed-point sum)."""
    text = ('{"width":%d,"height":%d,"samples_per_pixel":3,"max_depth":5,"sky":{"texture":""},"camera":{"look_from":{"x":0.0,"y":0.0,"z":0.0},'
            '"look_at":{"x":0.0,"y":0.0,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":90.0,"aspect":1.8},"objects":['
            '{"center":{"x":0.0,"y":0.0,"z":-1.0},"radius":0.5,"material":{"Lambertian":{"albedo":[0.8,0.3,0.3]}}}]}' % (w, h))
    sc = host.Scene.loads(text)
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    assert np.isnan(o_lin).all() and not o_rgb.any()
    for variant, pool in ((0, None), (1, None), (2, 0)):
        rgb, lin, st = gpu_render(sc, variant=variant, pool=pool)
        assert np.array_equal(rgb, o_rgb), (variant, rgb.ravel())
        assert st["segments"] == o_st["segments"] == w * h * 3
        if variant != 2:
            assert not lin.any()


def test_many_lights_nested_sampling(gpu_render, oracle, abi, host):
    """several lights + occluders: exercises the nested light-ray stack (raytracer.rs:103-110)"""
    objs = ['{"center":{"x":0.0,"y":-100.5,"z":-1.0},"radius":100.0,"material":{"Lambertian":{"albedo":[0.7,0.7,0.7]}}}']
    for i, x in enumerate((-2.0, 0.0, 2.0)):
        objs.append('{"center":{"x":%f,"y":2.5,"z":-2.0},"radius":0.5,"material":{"Light":{}}}' % x)
        objs.append('{"center":{"x":%f,"y":0.0,"z":-1.5},"radius":0.5,"material":{"%s}}' %
                    (x, ['Lambertian":{"albedo":[0.9,0.2,0.2]}', 'Glass":{"index_of_refraction":1.5}', 'Metal":{"albedo":[0.8,0.8,0.9],"fuzz":0.2}'][i]))
        objs.append('{"center":{"x":%f,"y":1.2,"z":-1.8},"radius":0.3,"material":{"Lambertian":{"albedo":[0.3,0.9,0.4]}}}' % x)
    text = ('{"width":64,"height":40,"samples_per_pixel":16,"max_depth":6,"sky":null,"camera":{"look_from":{"x":0.0,"y":1.0,"z":3.0},'
            '"look_at":{"x":0.0,"y":0.5,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":60.0,"aspect":1.6},"objects":[' + ",".join(objs) + "]}")
    sc = host.Scene.loads(text)
    assert len(sc.lights()) == 3
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    for variant, pool in ((0, None), (2, 0)):
        rgb, lin, st = gpu_render(sc, variant=variant, pool=pool)
        assert_parity(rgb, lin, o_rgb, o_lin, f"3 lights variant {variant}", atol=2e-6 if variant else pooled_atol(16))
        assert o_lin.max() > 0.05 and st["segments"] > st["samples"]


def test_procedural_10k_spheres(gpu_render, oracle, abi, host):
    """BASELINE configs[4] world (~10 000 spheres) at a size the oracle finishes in seconds"""
    import procedural
    sc = host.Scene.loads(procedural.make_json(width=64, height=36, spp=2, half=50, seed=0))
    assert sc.c.n_spheres == 10001
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    for variant, pool in ((0, None), (2, 0)):
        rgb, lin, st = gpu_render(sc, variant=variant, pool=pool)
        assert_parity(rgb, lin, o_rgb, o_lin, f"10k spheres variant {variant}")
        assert st["segments"] == o_st["segments"]
        if variant == 0:  # the grid does its job: a handful of exact tests per segment instead of 10 001
            assert st["grid_steps"] > 0 and st["exact_tests"] < 40 * st["segments"]


def test_host_buffer_entry_point(pkg, gpu_render, load_scene):
    """rt_render_rgb8 (host buffers in/out, the drop-in for render()'s loop) == device API"""
    sc = load_scene("cover", 80, 50, 2, 50)
    rgb, _, _ = gpu_render(sc)
    out, st = pkg.hip.render_rgb8(sc.ptr)
    assert np.array_equal(out, rgb) and st["kernel_ms"] > 0 and st["frame_ms"] >= st["kernel_ms"]


def test_cli_contract(tmp_path, gpu_render, load_scene):
    """main.rs:7-20: argv, the two stdout lines, PNG output; usage line + exit 0 on bad argc"""
    from PIL import Image
    exe = os.path.join(ROOT, "rust-raytracer_amd", "raytracer")
    cfg = json.load(open(os.path.join(ROOT, "scenes", "cfg1_test_800x600_spp16.json")))
    cfg.update(width=64, height=48, samples_per_pixel=4)
    p = tmp_path / "s.json"
    p.write_text(json.dumps(cfg))
    out = tmp_path / "o.png"
    r = subprocess.run([exe, str(p), str(out)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert lines[0] == "" and lines[1] == f"Rendering {out}" and lines[2].startswith("Frame time: ") and lines[2].endswith("ms")
    sc = load_scene(str(p))
    rgb, _, _ = gpu_render(sc)
    assert np.array_equal(np.asarray(Image.open(out)), rgb)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Usage: ") and "<config_file> <output_file>" in r.stdout
    r = subprocess.run([exe, "/nonexistent.json", str(out)], capture_output=True, text=True)
    assert r.returncode == 101 and "Unable to read config file." in r.stderr


def test_animation_driver(tmp_path, pkg, host, load_scene):
    """`raytracer <config> <prefix> --frames N --orbit DEG` (the reference's anim/frame_%03d.png
    workflow, README.md:43-57, main.rs:17): the scene stays resident, only the camera moves; every
    frame equals a render of the resident scene with that camera through the C ABI."""
    import math
    from PIL import Image
    exe = os.path.join(ROOT, "rust-raytracer_amd", "raytracer")
    cfg = json.load(open(os.path.join(ROOT, "scenes", "cfg1_test_800x600_spp16.json")))
    cfg.update(width=72, height=54, samples_per_pixel=3)
    p = tmp_path / "s.json"
    p.write_text(json.dumps(cfg))
    prefix = tmp_path / "frame"
    r = subprocess.run([exe, str(p), str(prefix), "--frames", "3", "--orbit", "25"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("\nRendering ") == 3 and r.stdout.count("Frame time: ") == 3
    sc = load_scene(str(p))
    gs = pkg.hip.HipScene(sc.ptr, 0)
    cam = (C.c_double * 11)()
    host.lib().rt_scene_camera.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    host.lib().rt_scene_camera.restype = None
    host.lib().rt_scene_camera(sc._h, cam)
    lf, la, up = list(cam[0:3]), list(cam[3:6]), list(cam[6:9])
    kl = math.sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2])
    k = [u / kl for u in up]
    frames = []
    for f in range(3):
        th = 25.0 * f * (3.14159265358979323846264338327950288 / 180.0)
        c, s_ = math.cos(th), math.sin(th)
        v = [lf[i] - la[i] for i in range(3)]
        kv = k[0] * v[0] + k[1] * v[1] + k[2] * v[2]
        kx = [k[1] * v[2] - k[2] * v[1], k[2] * v[0] - k[0] * v[2], k[0] * v[1] - k[1] * v[0]]
        frm = [la[i] + v[i] * c + kx[i] * s_ + k[i] * kv * (1.0 - c) for i in range(3)]
        out = (C.c_double * 13)()
        host.lib().rt_camera_derive((C.c_double * 3)(*frm), (C.c_double * 3)(*la), (C.c_double * 3)(*up), cam[9], cam[10], out)
        gs.set_camera(out[0:3], out[3:6], out[6:9], out[9:12])
        rgb, st = gs.render_to_host()
        assert st["samples"] == 72 * 54 * 3
        got = np.asarray(Image.open(f"{prefix}_{f:03d}.png"))
        assert np.array_equal(got, rgb), f"frame {f}"
        frames.append(rgb)
    gs.close()
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[1], frames[2])  # the camera did move


def test_full_size_headline_config_properties(gpu_render, oracle, abi, load_scene):
    """BASELINE configs[1] at FULL size (1200x800, spp 128, depth 50, 484 spheres), checked
    through size-independent properties: determinism, shard invariance, counter identities,
    and exact scanlines against the oracle."""
    sc = load_scene("cover")
    c = sc.c
    assert (c.width, c.height, c.samples_per_pixel, c.max_depth, c.n_spheres) == (1200, 800, 128, 50, 484)
    rgb, lin, st = gpu_render(sc)
    print(f"full config kernel_ms={st['kernel_ms']:.2f} segments/sample={st['segments'] / st['samples']:.3f} "
          f"exact tests/segment={st['exact_tests'] / st['segments']:.2f}")
    assert st["samples"] == 1200 * 800 * 128 and st["sphere_tests"] == st["segments"] * 484
    assert 2.0 < st["segments"] / st["samples"] < 3.5  # SURVEY §8d measured ~2.66
    rgb2, lin2, st2 = gpu_render(sc)
    assert np.array_equal(rgb, rgb2) and np.array_equal(lin, lin2) and st2["segments"] == st["segments"]
    # shard invariance: rank 3 of 8 renders exactly its scanlines of the full frame
    t = abi.RtRowTiles(8, 3, 8)
    rows = abi.tiles_global_rows(800, t)
    s_rgb, s_lin, _ = gpu_render(sc, tiles=t)
    assert np.array_equal(s_rgb, rgb[rows]) and np.array_equal(s_lin, lin[rows])
    # exact scanlines vs the oracle at full spp (tile {1 row, first y, stride huge} = one row)
    r_rgb, r_lin, r_st = gpu_render(sc, variant=2, pool=0)  # round-1 kernel, reference summation order: tight tolerance
    assert r_st["segments"] == st["segments"]
    for y in (5, 333, 640, 799):
        o_rgb, o_lin, _ = oracle.render(abi, sc.ptr, abi.RtRowTiles(1, y, 1 << 20))
        assert_parity(r_rgb[y:y + 1], r_lin[y:y + 1], o_rgb, o_lin, f"row {y}")
        assert_parity(rgb[y:y + 1], lin[y:y + 1], o_rgb, o_lin, f"row {y} pooled", atol=pooled_atol(128), flip_frac=2e-3)
    # image statistics sanity: top rows are sky gradient, bottom rows ground
    assert lin[:40].mean() > lin[-40:].mean()


@pytest.mark.parametrize("force_rccl", [False, True])
def test_bench_line_contract(force_rccl):
    """bench.py prints ONE JSON line with the driver's keys (+ roofline / cpu_baseline objects) and nothing
    else on stdout — also when an RCCL group is up (RT_BENCH_FORCE_COLLECTIVE: the N > 1 code path with one
    rank; RCCL's version banner must not reach stdout)."""
    env = dict(os.environ, MASTER_PORT="29561")
    if force_rccl:
        env["RT_BENCH_FORCE_COLLECTIVE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-row-stride", "200"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Msamples/s" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "BASELINE configs[1]" in d["config"]["workload"]
    samples = 1200 * 800 * 128
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * d["value"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("kind", range(6))
def test_fuzz_worlds_grid_equals_bruteforce_on_the_gpu(gpu_render, hostsim, host, kind):
    """The product kernel (grid walk) against the same kernel running the reference's scan over
    every sphere (variant 1): same bits, same path count — and the same bits as the CPU build of
    the lane logic."""
    rng = np.random.default_rng(1000 + kind)
    sc = host.Scene.loads(fuzz_world_json(rng, kind))
    a_rgb, a_lin, a_st = gpu_render(sc, variant=0)
    b_rgb, b_lin, b_st = gpu_render(sc, variant=1)
    assert a_st["segments"] == b_st["segments"], kind
    assert np.array_equal(a_lin, b_lin) and np.array_equal(a_rgb, b_rgb), kind
    h_rgb, h_lin, h_st = hostsim.render(sc.ptr, mode=3 + 16)
    assert int(h_st["segments"]) == a_st["segments"] and np.array_equal(h_lin, a_lin) and np.array_equal(h_rgb, a_rgb), kind
    if kind != 2:
        assert a_st["grid_steps"] > 0, "the world is expected to be gridded"


_RCCL_ONE_RANK = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # nccl == RCCL on ROCm
pkg = graft.load_package()
from rust_raytracer_amd import dist as rdist
sc = pkg.host.Scene.load("scenes/cfg2_cover_1200x800_spp128.json")
sc.c.width, sc.c.height, sc.c.samples_per_pixel = 96, 40, 2
W, H = 96, 40
stream = torch.cuda.current_stream()
pipe = rdist.FramePipeline(H, W, 0, 1, dev, force_collective=True)
frames, want = [], []
for i in range(5):
    gs = pkg.hip.HipScene(sc.ptr, 0)
    gs.set_option("seed", i)                       # every frame differs
    ref = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    gs.render(ref.data_ptr(), 0, None, stream.cuda_stream); gs.wait()
    want.append(ref.cpu().numpy())
    buf, done = pipe.begin(i)
    if done is not None:
        frames.append(done.cpu().numpy())
    gs.render(buf.data_ptr(), 0, None, stream.cuda_stream)
    pipe.submit(i)                                 # asynchronous gather over RCCL
    gs.wait(); gs.close()
frames += [f.cpu().numpy() for f in pipe.drain()]
dist.barrier(); torch.cuda.synchronize()
assert len(frames) == 5
for i in range(5):
    assert np.array_equal(frames[i], want[i]), i
assert not np.array_equal(want[0], want[1])
dist.destroy_process_group()
print("RCCL_PIPELINE_OK")
'''


def test_frame_pipeline_through_rccl_one_rank(tmp_path):
    """bench.py's N > 1 machinery (double-buffered tiles, asynchronous `gather` on RCCL's stream,
    row permutation on the destination) with the only process group a 1-GPU box allows: one rank."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    r = subprocess.run([sys.executable, str(script), str(port)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "RCCL_PIPELINE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_device_sphere_hit_matches_the_host_build(pkg, hostsim, abi, torch_cuda):
    """Sphere::hit pair by pair on the device (the kernel's own exact_hit_any_order) against the CPU
    build of the same source: random pairs, rays tangent to the sphere (discriminant exactly 0 or in
    the denormal range — the cold library-sqrt path), origins on / inside the sphere, negative
    radii, degenerate directions."""
    torch = torch_cuda
    rng = np.random.default_rng(99)
    n = 200_000
    c = rng.uniform(-5, 5, (n, 3))
    r = rng.uniform(0.05, 3.0, n) * rng.choice([1.0, 1.0, 1.0, -1.0], n)
    o = c + rng.standard_normal((n, 3)) * rng.uniform(0.0, 6.0, (n, 1))
    d = (c + rng.standard_normal((n, 3)) * np.abs(r)[:, None] * rng.uniform(0.0, 1.5, (n, 1))) - o
    k = n // 8
    # exactly tangent, representable: axis-aligned rays past integer-ish spheres  (oc.d)^2 - |d|^2 (|oc|^2 - r^2) = 0
    ci = rng.integers(-4, 5, (k, 3)).astype(np.float64); ri = rng.integers(1, 4, k).astype(np.float64)
    ax = rng.integers(0, 3, k); bx = (ax + 1) % 3
    ot = ci.copy(); ot[np.arange(k), bx] += ri; ot[np.arange(k), ax] -= rng.integers(2, 9, k)
    dt = np.zeros((k, 3)); dt[np.arange(k), ax] = rng.choice([0.5, 1.0, 2.0, 4.0], k)
    c[:k], r[:k], o[:k], d[:k] = ci, ri, ot, dt
    # almost tangent: the same rays nudged by a few ulps / tiny offsets either way (tiny positive and negative discriminants)
    c[k:2 * k], r[k:2 * k], d[k:2 * k] = ci, ri, dt
    o[k:2 * k] = ot
    o[np.arange(k, 2 * k), bx] += ri * rng.choice([-1.0, 1.0], k) * 10.0 ** rng.uniform(-150, -17, k)
    # origin exactly on the surface / inside; zero and denormal direction components
    o[2 * k:3 * k] = c[2 * k:3 * k] + np.eye(3)[rng.integers(0, 3, k)] * np.abs(r[2 * k:3 * k, None])
    d[3 * k:3 * k + k // 2, rng.integers(0, 3)] = rng.choice([0.0, -0.0, 1e-310, -1e-305], k // 2)
    rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
    sph = np.ascontiguousarray(np.concatenate([c, r[:, None]], axis=1))
    d_rays, d_sph = torch.from_numpy(rays).cuda(), torch.from_numpy(sph).cuda()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    assert pkg.hip.lib().rt_hip_hit_probe(d_rays.data_ptr(), d_sph.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = np.empty(n)
    s = abi.RtSphere()
    tmax = float(np.finfo(np.float64).max)
    for i in range(n):
        s.center[:] = c[i].tolist(); s.radius = float(r[i])
        want[i] = hostsim.hostsim_exact_root(dvec(*o[i]), dvec(*d[i]), s, 0.001, tmax)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert (want[:k] > 0).mean() > 0.9 and (want >= 0).mean() > 0.3   # the tangent rays do hit (discriminant 0 -> one root)
