"""CPU-side checks of the kernel's per-lane logic (rt_core.h compiled for the host by
tests/hostsim — a development tool, not a product path) against the oracle, plus the
oracle against its frozen golden vectors.  The real parity tests run on the GPU
(tests/test_gpu_parity.py)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from conftest import dvec
from parity import assert_parity

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {
    "cover_96x64_spp4": ("cover", 96, 64, 4, 50, 0),
    "cover_60x40_spp2_seed7": ("cover", 60, 40, 2, 50, 7),
    "test_80x60_spp4": ("test", 80, 60, 4, 8, 0),
    "test_40x30_spp8_depth50": ("test", 40, 30, 8, 50, 3),
    "cover_tex_64x36_spp4": ("cover4k_tex", 64, 36, 4, 50, 0),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name, oracle, abi, load_scene):
    scene, w, h, spp, depth, seed = CASES[name]
    sc = load_scene(scene, w, h, spp, depth, seed)
    rgb, lin, st = oracle.render(abi, sc.ptr)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert st["samples"] == int(g["samples"]) == w * h * spp
    # every operation of the oracle is IEEE (atan2 included: the shared double-double routine): bit-exact anywhere
    assert st["segments"] == int(g["segments"]) and st["segments_discarded"] == int(g["segments_discarded"])
    assert np.array_equal(rgb, g["rgb8"]) and np.array_equal(lin, g["linear"])
    assert (st["segments_discarded"] > 0) == (name.startswith("test_"))   # only the lit scene has light loops to discard


@pytest.mark.parametrize("name", sorted(CASES))
def test_core_matches_oracle(name, hostsim, oracle, abi, load_scene):
    scene, w, h, spp, depth, seed = CASES[name]
    sc = load_scene(scene, w, h, spp, depth, seed)
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    images = []
    # product behaviour (grid walk + exact test), round-1 kernel (cull + exact confirm), exact test on every sphere
    for mode in (3, 1, 0):
        rgb, lin, st = hostsim.render(sc.ptr, None, mode)
        assert_parity(rgb, lin, o_rgb, o_lin, f"{name} mode {mode}")
        assert st["samples"] == o_st["samples"]
        # the kernel skips exactly the light loops the reference computes and discards (raytracer.rs:124)
        assert st["segments"] == o_st["segments"] - o_st["segments_discarded"]
        images.append((rgb, lin))
    assert st["exact_tests"] == st["sphere_tests"]
    for rgb, lin in images[1:]:  # hit_world variants pick the same (t, sphere) for every ray: identical bits
        assert np.array_equal(rgb, images[0][0]) and np.array_equal(lin, images[0][1])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_adversarial_sphere_records(hostsim, oracle, abi, host, seed):
    """NaN / inf / 1e300 / denormal / zero radii and NaN / inf / 1e308 centres among ordinary spheres (records only the C
    ABI can carry): the lane logic with the grid walk, with brute force, and the oracle agree — same NaN pixels, same
    bits elsewhere; per segment the walk and the object-order scan pick the same (t, sphere)."""
    from fuzz_worlds import adversarial_scene
    sc = adversarial_scene(host, seed)
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    nan = np.isnan(o_lin)
    out = []
    for mode in (3, 0):
        rgb, lin, st = hostsim.render(sc.ptr, None, mode)
        assert np.array_equal(np.isnan(lin), nan), mode
        assert np.abs(np.where(nan, 0.0, lin) - np.where(nan, 0.0, o_lin)).max() <= 1e-6, mode
        assert np.abs(rgb.astype(int) - o_rgb.astype(int)).max() <= 1
        assert st["segments"] == o_st["segments"] - o_st["segments_discarded"]
        out.append((rgb, np.where(nan, 0.0, lin)))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    _, _, st = hostsim.render(sc.ptr, None, 4)
    assert st["kernel_ms"] == 0.0 and st["grid_steps"] > 0


def _many_lights_scene(host, n_lights=3, w=64, h=40, spp=16, depth=6):
    objs = ['{"center":{"x":0.0,"y":-100.5,"z":-1.0},"radius":100.0,"material":{"Lambertian":{"albedo":[0.7,0.7,0.7]}}}']
    for i, x in enumerate((-2.0, 0.0, 2.0)[:n_lights]):
        objs.append('{"center":{"x":%f,"y":2.5,"z":-2.0},"radius":0.5,"material":{"Light":{}}}' % x)
        objs.append('{"center":{"x":%f,"y":0.0,"z":-1.5},"radius":0.5,"material":{"%s}}' %
                    (x, ['Lambertian":{"albedo":[0.9,0.2,0.2]}', 'Glass":{"index_of_refraction":1.5}', 'Metal":{"albedo":[0.8,0.8,0.9],"fuzz":0.2}'][i]))
        objs.append('{"center":{"x":%f,"y":1.2,"z":-1.8},"radius":0.3,"material":{"Lambertian":{"albedo":[1.0,0.9,0.4]}}}' % x)
    text = ('{"width":%d,"height":%d,"samples_per_pixel":%d,"max_depth":%d,"sky":{"texture":""},"camera":{"look_from":{"x":0.0,"y":1.0,"z":3.0},'
            '"look_at":{"x":0.0,"y":0.5,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":60.0,"aspect":1.6},"objects":[' % (w, h, spp, depth) + ",".join(objs) + "]}")
    return host.Scene.loads(text)


def test_short_colour_maps_are_bit_identical_to_the_general_one(hostsim, oracle, abi, host, load_scene):
    """Scenes whose albedos all lie in [0, 1] run the kernel's SHORT colour maps (rt_core.h): unlit G(x) = q x; lit
    G(x) = min(h, p + q x) with q in registers and (p, h) in memory once a light contributed.  Same bits as the general
    clamped-affine map FwdT<false> — on the reference's lit test_scene, on a gradient-sky world with three lights whose
    sums hit the clamp (levels 0 AND 1 sampling the lights, nested light rays), on the unlit cover scene — and the
    oracle's image within the parity bar."""
    cases = [load_scene("test", 96, 72, 8, 8), load_scene("test", 64, 48, 8, 50, 5), _many_lights_scene(host), _many_lights_scene(host, 2, 48, 30, 32, 3),
             _many_lights_scene(host, 1, 48, 30, 16, 2), load_scene("cover", 96, 64, 4, 50)]
    for i, sc in enumerate(cases):
        g_rgb, g_lin, g_st = hostsim.render(sc.ptr, None, 3 + 16)        # general map
        s_rgb, s_lin, s_st = hostsim.render(sc.ptr, None, 3 + 16 + 32)   # short maps
        assert np.array_equal(g_lin, s_lin) and np.array_equal(g_rgb, s_rgb), f"case {i}: {int((g_lin != s_lin).sum())} values differ"
        assert g_st["segments"] == s_st["segments"]
        o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
        assert_parity(s_rgb, s_lin, o_rgb, o_lin, f"short map case {i}", atol=2e-6 + 3e-8 * sc.c.samples_per_pixel, flip_frac=2e-3)
        if len(sc.lights()):
            assert o_st["segments_discarded"] > 0 and s_lin.max() > 0.5


def test_division_by_255_through_the_reciprocal_is_the_ieee_quotient(hostsim):
    """rt_div255f (texel / sky-texel colours, materials.rs:247-252, raytracer.rs:153-158): one Markstein correction on
    x * RN(1/255) equals x / 255.0f — on every value the kernel divides (bytes, 0.7f * bytes) and on 2e7 floats spread over
    [0, 256] incl. denormals (tools/analysis/div255_check.cpp runs all 1.13e9)."""
    rng = np.random.default_rng(3)
    b = np.arange(256, dtype=np.float32)
    bits = rng.integers(0, np.float32(256.0).view(np.uint32) + 1, 20_000_000, dtype=np.uint32)
    x = np.concatenate([b, np.float32(0.7) * b, bits.view(np.float32), np.array([0.0, 1e-45, 1e-40, 255.0, 256.0], np.float32)]).astype(np.float32)
    out = np.zeros_like(x)
    hostsim.hostsim_div255(x.ctypes.data, out.ctypes.data, len(x))
    want = x / np.float32(255.0)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), int((out != want).sum())


def test_fused_exact_conversions_equal_the_rounded_forms(hostsim):
    """range_m1_1 as ONE fma (w * 2^-31 - 1) against the oracle's three rounded operations ((w * 2^-32) * 2 + (-1): every
    intermediate is exact), u01_53 from two exact 32-bit conversions against the 64-bit integer conversion, and
    sample_to_fixed's fused x * 2^40 + 0.5 against multiply-then-add (the product is exact)."""
    rng = np.random.default_rng(5)
    w = np.concatenate([rng.integers(0, 1 << 32, 8_000_000, dtype=np.uint64).astype(np.uint32),
                        np.array([0, 1, 2, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFE, 0xFFFFFFFF], np.uint32)])
    got = np.zeros(len(w), np.float64)
    hostsim.hostsim_range_m1_1(w.ctypes.data, got.ctypes.data, len(w))
    want = (w.astype(np.float64) * (1.0 / 4294967296.0)) * 2.0 + (-1.0)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    lo = np.concatenate([rng.integers(0, 1 << 32, 4_000_000, dtype=np.uint64).astype(np.uint32), np.array([0, 0xFFFFFFFF, 0x7FF, 0x800, 0xFFFFFFFF, 0], np.uint32)])
    hi = np.concatenate([rng.integers(0, 1 << 32, 4_000_000, dtype=np.uint64).astype(np.uint32), np.array([0, 0xFFFFFFFF, 0, 0, 0x7FF, 0x800], np.uint32)])
    got = np.zeros(len(lo), np.float64)
    hostsim.hostsim_u01_53(lo.ctypes.data, hi.ctypes.data, got.ctypes.data, len(lo))
    u = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    want = (u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)      # rand 0.8's form (u64 -> f64 is exact below 2^53)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) and got.max() < 1.0
    v = np.concatenate([rng.random(4_000_000).astype(np.float32), rng.integers(0, 0x3F800001, 4_000_000, dtype=np.uint32).view(np.float32),
                        np.array([0.0, 1.0, 1e-45, 9.09e-13, 4.5e-13, 0.5, np.nextafter(np.float32(1), np.float32(0))], np.float32)]).astype(np.float32)
    got = np.zeros(len(v), np.uint64)
    hostsim.hostsim_sample_to_fixed(v.ctypes.data, got.ctypes.data, len(v))
    want = np.trunc(v.astype(np.float64) * 1099511627776.0 + 0.5).astype(np.uint64)
    assert np.array_equal(got, want)


def test_fast_texel_path_agrees_with_the_exact_one(hostsim):
    """texel_fast (plain-f64 unit vector and atan, rt_core.h) names a texel only when the exact path (correctly-rounded
    divisions and atan2, then floor) names the same one; hit points aimed at texel boundaries, at the u wrap (rot = 1),
    at the poles and at the octant seams of the atan reduction either agree or are handed to the exact path."""
    import texel_points
    rng = np.random.default_rng(11)
    total = wrong = refused = 0
    for (w, h, h_off, radius, centre) in texel_points.CASES:
        n = int(os.environ.get("RT_TEXEL_POINTS", "400000"))
        pts, m4 = texel_points.points(rng, n, w, h, h_off, radius, centre)
        out = np.zeros((n, 5), np.uint64)
        hostsim.hostsim_texels(pts.ctypes.data, n, dvec(*centre, radius), h_off, w, h, out.ctypes.data)
        ok = out[:, 0] == 1
        wrong += int((ok & ((out[:, 1] != out[:, 3]) | (out[:, 2] != out[:, 4]))).sum())
        refused += int((~ok[m4:]).sum())   # of the un-aimed, uniformly distributed directions
        total += n - m4
    assert wrong == 0, f"{wrong} of {total} fast texels differ from the exact path"
    # ... and it takes practically every ordinary hit point (a boundary band is 4e-12 of a texel wide)
    assert refused < 1e-4 * total, (refused, total)


def test_grid_walk_matches_brute_force_per_segment(hostsim, load_scene, host):
    """audit mode 4: for EVERY ray segment of these renders, the grid walk and the reference's
    object-order scan over all spheres (raytracer.rs:52-57) return the same (t, sphere) bits."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "scenes"))
    import procedural
    cases = [load_scene("cover", 150, 100, 4, 50), load_scene("test", 80, 60, 4, 8), load_scene("cover4k_tex", 64, 36, 2, 50),
             host.Scene.loads(procedural.make_json(width=48, height=27, spp=2, half=50, seed=0)),
             host.Scene.loads(procedural.make_json(width=64, height=36, spp=2, half=50, seed=0))]  # (once caught a ray grazing the grid's top face)
    for sc in cases:
        _, _, st = hostsim.render(sc.ptr, None, 4)
        assert st["kernel_ms"] == 0.0, f"{st['kernel_ms']} segments where the grid walk disagrees with brute force"
        if sc.c.n_spheres > 64:  # and it actually prunes
            assert st["grid_steps"] > 0 and st["exact_tests"] < 0.05 * st["sphere_tests"]
        else:
            assert st["grid_steps"] == 0 and st["exact_tests"] == st["sphere_tests"]


def test_cull_never_rejects_a_hit_in_scenes(hostsim, load_scene):
    """audit mode: every sphere gets the exact test AND the cull; false rejects must be 0"""
    for scene, w, h, spp, depth in (("cover", 120, 80, 4, 50), ("test", 80, 60, 4, 8), ("cover4k_tex", 64, 36, 2, 50)):
        sc = load_scene(scene, w, h, spp, depth)
        _, _, st = hostsim.render(sc.ptr, None, 2)
        assert st["kernel_ms"] == 0.0, f"{scene}: {st['kernel_ms']} false rejects"
        _, _, st1 = hostsim.render(sc.ptr, None, 1)
        assert st1["exact_tests"] < 0.02 * st1["sphere_tests"] or sc.c.n_spheres < 16  # and it actually culls


def test_cull_margin_adversarial(hostsim, oracle, abi):
    """Near-tangent rays at many scales and offsets from the origin: whenever the exact f64
    Sphere::hit (sphere.rs:46-58) accepts, the f32 cull must pass (disc >= 0 or NaN)."""
    rng = np.random.default_rng(1234)
    L = oracle.lib(abi)
    out = (C.c_double * 10)()
    n_hit = n_checked = 0
    for trial in range(40000):
        scale = 10.0 ** rng.uniform(-3, 3.5)
        offs = 10.0 ** rng.uniform(-2, 4) * rng.standard_normal(3) if trial % 3 else np.zeros(3)
        c = offs + scale * rng.standard_normal(3)
        r = scale * 10.0 ** rng.uniform(-2, 1) * (1 if trial % 7 else -1)
        n = rng.standard_normal(3); n /= np.linalg.norm(n)
        tdir = np.cross(n, rng.standard_normal(3)); tdir /= np.linalg.norm(tdir)
        grazing = abs(r) * (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-16, -3))
        p = c + n * grazing
        dist = abs(r) * 10.0 ** rng.uniform(-2, 3)
        dlen = 10.0 ** rng.uniform(-3, 3)
        o = p - tdir * dist
        d = tdir * dlen
        s = abi.RtSphere(radius=r)
        for i in range(3):
            s.center[i] = c[i]
        hit = L.rt_oracle_sphere_hit(dvec(*c), r, dvec(*o), dvec(*d), 0.001, math.inf, out)
        disc = hostsim.hostsim_cull_disc(dvec(*o), dvec(*d), C.byref(s))
        n_checked += 1
        if hit:
            n_hit += 1
            assert not (disc < 0.0), (trial, disc, c, r, o, d)
    assert n_hit > n_checked // 10


def test_exact_root_equals_sphere_hit(hostsim, oracle, abi):
    """the kernel's root selection (incl. its exact behind-the-ray shortcut) returns exactly the
    t that the oracle's Sphere::hit accepts, for origins inside/outside/behind and any t_max"""
    rng = np.random.default_rng(99)
    L = oracle.lib(abi)
    out = (C.c_double * 10)()
    hits = culled = 0
    for trial in range(30000):
        c = rng.standard_normal(3) * 10.0 ** rng.uniform(-1, 2)
        r = 10.0 ** rng.uniform(-2, 2) * (1 if trial % 5 else -1)
        o = c + rng.standard_normal(3) * abs(r) * 10.0 ** rng.uniform(-1, 1)
        if trial % 4 == 0:  # start exactly on the surface-ish (bounced rays)
            n = rng.standard_normal(3); n /= np.linalg.norm(n)
            o = c + n * abs(r) * (1 + rng.uniform(-1e-12, 1e-12))
        d = rng.standard_normal(3) * 10.0 ** rng.uniform(-2, 2)
        t_max = math.inf if trial % 3 else 10.0 ** rng.uniform(-2, 2)
        s = abi.RtSphere(radius=r)
        for i in range(3):
            s.center[i] = c[i]
        hit = L.rt_oracle_sphere_hit(dvec(*c), r, dvec(*o), dvec(*d), 0.001, t_max, out)
        got = hostsim.hostsim_exact_root(dvec(*o), dvec(*d), C.byref(s), 0.001, min(t_max, 1.7976931348623157e308))
        may = hostsim.hostsim_hit_prefix(dvec(*o), dvec(*d), C.byref(s))   # the prefix alone never rejects a sphere the test accepts
        if hit:
            hits += 1
            assert got == out[0], (trial, got, out[0])
            assert may == 1, (trial, "prefix rejected an accepted hit")
        else:
            assert got < 0.0, (trial, got)
        culled += may == 0
    assert hits > 5000 and culled > 5000


def test_row_tiles_are_bit_identical(hostsim, oracle, abi, load_scene):
    """RNG is addressed by global pixel index: any tiling reproduces the full frame exactly."""
    sc = load_scene("cover", 40, 27, 2, 50)
    full_rgb, full_lin, _ = oracle.render(abi, sc.ptr)
    for world in (2, 3):
        for rank in range(world):
            t = abi.RtRowTiles(8, rank, world)
            rows = abi.tiles_global_rows(27, t)
            rgb, lin, _ = oracle.render(abi, sc.ptr, t)
            assert np.array_equal(rgb, full_rgb[rows]) and np.array_equal(lin, full_lin[rows])
            rgb2, lin2, _ = hostsim.render(sc.ptr, t, 1)
            assert_parity(rgb2, lin2, full_rgb[rows], full_lin[rows])


def test_degenerate_scenes(hostsim, oracle, abi, host):
    """edge cases: empty world, max_depth 0/1, single pixel column, null sky, light-only"""
    base = ('{"width":9,"height":5,"samples_per_pixel":3,"max_depth":%d,"sky":%s,"camera":{"look_from":{"x":0.0,"y":0.0,"z":0.0},'
            '"look_at":{"x":0.0,"y":0.0,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":90.0,"aspect":1.8},"objects":[%s]}')
    lam = '{"center":{"x":0.0,"y":0.0,"z":-1.0},"radius":0.5,"material":{"Lambertian":{"albedo":[0.8,0.3,0.3]}}}'
    light = '{"center":{"x":0.0,"y":3.0,"z":-1.0},"radius":1.0,"material":{"Light":{}}}'
    bright = '{"center":{"x":0.0,"y":-100.5,"z":-1.0},"radius":100.0,"material":{"Metal":{"albedo":[1.5,0.2,2.0],"fuzz":0.3}}}'
    for depth in (0, 1, 2, 5):
        for sky in ("null", '{"texture":""}'):
            for objs in ("", lam, lam + "," + light, light + "," + lam + "," + light + "," + bright):
                sc = host.Scene.loads(base % (depth, sky, objs))
                o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
                rgb, lin, st = hostsim.render(sc.ptr, None, 1)
                assert_parity(rgb, lin, o_rgb, o_lin, f"depth {depth} sky {sky} objs {len(objs)}")


def _random_scene(abi, rng, n, spread, r_lo, r_hi, big=None):
    spheres = (abi.RtSphere * (n + (1 if big else 0)))()
    for i in range(n):
        c = rng.uniform(-spread, spread, 3)
        spheres[i].center[:] = list(c)
        spheres[i].radius = float(rng.uniform(r_lo, r_hi)) * (-1.0 if i % 11 == 0 else 1.0)
        spheres[i].kind = abi.RT_MAT_LAMBERTIAN
    if big:
        spheres[n].center[:] = [0.0, -big - spread, 0.0]
        spheres[n].radius = big
    sc = abi.RtScene(abi_version=abi.RT_ABI_VERSION, width=4, height=4, samples_per_pixel=1, max_depth=2,
                     spheres=spheres, n_spheres=len(spheres))
    return sc, spheres


def test_grid_walk_adversarial_rays(hostsim, abi):
    """Random worlds (dense, sparse, flat, with a huge ground sphere, far from the origin) and
    rays chosen to stress the walk: origins on sphere surfaces, grazing tangents, axis-parallel
    and cell-boundary-aligned directions, origins far outside the grid.  The grid walk must
    return exactly the brute-force (t, sphere) every time."""
    rng = np.random.default_rng(77)
    out = (C.c_int * 2)()
    t_out = (C.c_double * 2)()
    n_rays = n_hits = 0
    worlds = [dict(n=200, spread=5.0, r_lo=0.05, r_hi=0.6, big=None), dict(n=600, spread=20.0, r_lo=0.1, r_hi=0.3, big=1000.0),
              dict(n=80, spread=1.0, r_lo=0.2, r_hi=0.5, big=None), dict(n=300, spread=8.0, r_lo=0.01, r_hi=2.5, big=None),
              dict(n=900, spread=25.0, r_lo=0.2, r_hi=0.2, big=1000.0)]
    for wi, wd in enumerate(worlds):
        sc, spheres = _random_scene(abi, rng, **wd)
        if wi == 1:  # flat world: every centre near y = 0
            for i in range(wd["n"]):
                spheres[i].center[1] = float(rng.uniform(0.0, 0.3))
        if wi == 4:  # exactly one layer of equal spheres on the ground: the grid is a single cell high
            for i in range(wd["n"]):
                spheres[i].center[1] = 0.2
                spheres[i].radius = 0.2
        if wi == 3:  # far from the origin: large coordinates, small spheres
            for i in range(wd["n"]):
                for k in range(3):
                    spheres[i].center[k] += 5000.0
        info = (C.c_uint32 * 6)()
        assert hostsim.hostsim_grid_info(C.byref(sc), info) == 0 and info[0] > 0, "world must be gridded"
        n = wd["n"]
        for trial in range(1500):
            i = int(rng.integers(n))
            c = np.array(spheres[i].center[:]); r = abs(spheres[i].radius)
            kind = trial % 7
            nrm = rng.standard_normal(3); nrm /= np.linalg.norm(nrm)
            if kind == 0:    # leaves a sphere surface in a random direction (a bounced ray)
                o = c + nrm * r; d = rng.standard_normal(3)
            elif kind == 1:  # grazes sphere i: offset from the centre ~ r (1 +- tiny)
                tdir = np.cross(nrm, rng.standard_normal(3)); tdir /= np.linalg.norm(tdir)
                p = c + nrm * r * (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-15, -2))
                o = p - tdir * rng.uniform(0.5, 30.0); d = tdir * rng.uniform(0.1, 3.0)
            elif kind == 2:  # axis-parallel through the sphere's bounding box
                ax = int(rng.integers(3)); d = np.zeros(3); d[ax] = rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 2.0)
                o = c + rng.uniform(-1.2, 1.2, 3) * r; o[ax] -= np.sign(d[ax]) * rng.uniform(1.0, 40.0)
            elif kind == 3:  # from far outside the grid towards a sphere
                o = c + nrm * 10.0 ** rng.uniform(1, 4.5); d = (c + rng.uniform(-1, 1, 3) * r * 1.5) - o
            elif kind == 4:  # one direction component denormal / zero, the others diagonal
                d = rng.choice([-1.0, 1.0], 3); d[int(rng.integers(3))] = rng.choice([0.0, -0.0, 1e-310, -1e-300, 1e-40])
                o = c - d * rng.uniform(0.5, 10.0) + rng.uniform(-1, 1, 3) * r
            elif kind == 5:  # starts inside a sphere
                o = c + nrm * r * rng.uniform(0.0, 0.999); d = rng.standard_normal(3) * 10.0 ** rng.uniform(-3, 3)
            else:            # hits sphere i at (almost) its extreme point along an axis — for the outermost
                             # spheres that is on the grid's outer face — coming in nearly parallel to that face
                ax = int(rng.integers(3)); sgn = rng.choice([-1.0, 1.0])
                e = np.zeros(3); e[ax] = sgn
                p = c + e * r * (1.0 - 10.0 ** rng.uniform(-12, -2))
                tdir = rng.standard_normal(3); tdir[ax] = 0.0; tdir /= np.linalg.norm(tdir)
                d = tdir + e * 10.0 ** rng.uniform(-4, -1.5)
                o = p - d * rng.uniform(0.5, 12.0)
            assert hostsim.hostsim_hit_world(C.byref(sc), dvec(*o), dvec(*d), out, t_out) == 0
            n_rays += 1
            n_hits += out[1] >= 0
            assert out[0] == out[1] and (out[0] < 0 or t_out[0] == t_out[1]), (wi, trial, kind, out[:], t_out[:])
    assert n_hits > 0.3 * n_rays


def test_rays_from_far_away_walk_the_grid(hostsim, abi, monkeypatch):
    """A ray that comes back from the far side of the r = 1000 ground (a Glass sphere touching the ground refracts rays into
    it; they leave it 2 000 units away) enters the grid from ~10^4 cells away when the cells are thin.  The entry point is
    taken a fraction `slack` of the way back towards the origin; at 2^-12 (until round 5) such a ray landed more than two
    cells outside the grid and took the full scan over every sphere — correct, and 100 ms for ONE ray of a frame of a
    2 x 10^5-sphere world on the GPU.  At 2^-16 it walks — and finds what brute force finds."""
    rng = np.random.default_rng(17)
    sc, spheres = _random_scene(abi, rng, n=900, spread=25.0, r_lo=0.2, r_hi=0.2, big=1000.0)
    for i in range(900):
        spheres[i].center[1] = 0.2
        spheres[i].radius = 0.2
    monkeypatch.setenv("RT_GRID_N", "60,4,60")   # (tests/hostsim is built with -DRT_DEV_KNOBS: thin cells, 0.1 units high)
    info = (C.c_uint32 * 6)()
    assert hostsim.hostsim_grid_info(C.byref(sc), info) == 0 and info[1] == 4
    out = (C.c_int * 2)()
    t_out = (C.c_double * 2)()
    n_hit = 0
    for trial in range(600):
        i = int(rng.integers(900))
        c = np.array(spheres[i].center[:])
        # origin on the ground sphere's far side (y ~ -1000 .. -2000) or 10^3 .. 10^4 units away in any direction
        if trial % 2:
            a = rng.uniform(0.0, 0.45 * np.pi); phi = rng.uniform(0, 2 * np.pi)
            o = np.array([1000.0 * np.sin(a) * np.cos(phi), -1000.0 - 1000.0 * np.cos(a), 1000.0 * np.sin(a) * np.sin(phi)])
        else:
            nrm = rng.standard_normal(3); nrm /= np.linalg.norm(nrm)
            o = c + nrm * 10.0 ** rng.uniform(3, 4)
        d = (c + rng.uniform(-1, 1, 3) * 0.3) - o
        d *= 10.0 ** rng.uniform(-3, 1)
        assert hostsim.hostsim_grid_mode(C.byref(sc), dvec(*o), dvec(*d)) in (0, 1), "a far ray took the full scan"
        assert hostsim.hostsim_hit_world(C.byref(sc), dvec(*o), dvec(*d), out, t_out) == 0
        assert out[0] == out[1] and (out[0] < 0 or t_out[0] == t_out[1]), (trial, out[:], t_out[:])
        n_hit += out[1] >= 0
    assert n_hit > 100


def test_any_order_hit_equals_object_order_scan(hostsim, abi):
    """coincident and duplicated spheres: ties in t must go to the lowest object index
    (raytracer.rs:52-57 keeps the first of equals), whatever order the cells list them in"""
    rng = np.random.default_rng(5)
    sc, spheres = _random_scene(abi, rng, 120, 4.0, 0.2, 0.5)
    for i in range(0, 120, 3):  # exact duplicates and same-centre shells
        spheres[i + 1].center[:] = spheres[i].center[:]
        spheres[i + 1].radius = spheres[i].radius
        spheres[i + 2].center[:] = spheres[i].center[:]
    out = (C.c_int * 2)()
    t_out = (C.c_double * 2)()
    for trial in range(3000):
        i = int(rng.integers(120))
        c = np.array(spheres[i].center[:])
        o = c + rng.standard_normal(3) * 6.0
        d = (c + rng.standard_normal(3) * 0.2) - o
        assert hostsim.hostsim_hit_world(C.byref(sc), dvec(*o), dvec(*d), out, t_out) == 0
        assert out[0] == out[1] and (out[0] < 0 or t_out[0] == t_out[1]), (trial, out[:], t_out[:])


def test_div_by_recip_is_exact(hostsim):
    """rt_core.h div_by_recip (Markstein's division through RN(1/b), two FMA corrections) must
    equal the IEEE quotient bit for bit wherever the kernel uses it: random operands over the
    supported range, plus divisors whose significand is all ones / just above a power of two."""
    rng = np.random.default_rng(2024)
    n = 10_000_000

    def rnd(emin, emax, m):
        mant = rng.integers(0, 1 << 52, m, dtype=np.uint64)
        e = rng.integers(emin + 1023, emax + 1024, m, dtype=np.uint64)
        sign = rng.integers(0, 2, m, dtype=np.uint64) << np.uint64(63)
        return (sign | (e << np.uint64(52)) | mant).view(np.float64)

    x = np.concatenate([rnd(-20, 20, n // 2), rnd(-140, 140, n // 4), rnd(-3, 3, n // 4)])
    b = np.concatenate([rnd(-20, 20, n // 2), rnd(-140, 140, n // 4), rnd(-1, 1, n // 4)])
    special = (np.uint64(1023) << np.uint64(52)) | np.concatenate([
        (np.uint64((1 << 52) - 1) - rng.integers(0, 64, n // 8, dtype=np.uint64)), rng.integers(0, 64, n // 8, dtype=np.uint64)])
    b[-(n // 4):] = special.view(np.float64)
    out = np.empty_like(x)
    hostsim.hostsim_div_by_recip(x.ctypes.data, b.ctypes.data, out.ctypes.data, len(x))
    want = x / b
    bad = out != want
    assert not bad.any(), (int(bad.sum()), x[bad][:3], b[bad][:3])


def test_grid_policy_on_the_reference_scenes(hostsim, load_scene):
    """build_grid's choices on the scenes that matter: the cover scene keeps its ground and its
    three r = 1 spheres out of the grid (a flat one-cell-high grid around the 480 small spheres),
    the 7-sphere test scene is not gridded at all, every sphere is either gridded or `large`."""
    info = (C.c_uint32 * 6)()
    sc = load_scene("cover", 64, 48, 1, 50)
    assert hostsim.hostsim_grid_info(sc.ptr, info) == 0
    nx, ny, nz, n_large, n_cells, n_items = info[:]
    assert n_large == 4 and ny == 1 and 24 <= nx <= 96 and 24 <= nz <= 96
    assert n_cells == (nx + 2) * (ny + 2) * (nz + 2) and n_items >= 480
    sc = load_scene("test", 64, 48, 1, 8)
    assert hostsim.hostsim_grid_info(sc.ptr, info) == 0
    assert info[0] == 0 and info[3] == sc.c.n_spheres == 7


@pytest.mark.parametrize("kind", range(6))
def test_fuzz_worlds_grid_audit(hostsim, host, kind):
    """tests/fuzz_worlds.py through the per-segment audit (mode 4: every segment's grid-walk hit
    against the reference's scan over all spheres) — the same worlds the GPU test renders."""
    from fuzz_worlds import fuzz_world_json
    sc = host.Scene.loads(fuzz_world_json(np.random.default_rng(1000 + kind), kind))
    _, _, st = hostsim.render(sc.ptr, mode=4)
    assert st["kernel_ms"] == 0.0, f"{int(st['kernel_ms'])} segments where the grid walk and the brute-force scan disagree"
    assert st["segments"] > sc.c.width * sc.c.height * sc.c.samples_per_pixel
    a = hostsim.render(sc.ptr, mode=3 + 16)
    b = hostsim.render(sc.ptr, mode=0 + 16)
    assert np.array_equal(a[1], b[1]) and a[2]["segments"] == b[2]["segments"]


@pytest.mark.parametrize("kind", range(6))
def test_fuzz_worlds_lane_logic_matches_the_oracle(hostsim, oracle, abi, host, kind):
    """The per-lane code the kernel runs (CPU build, grid walk, fixed-point sums) against the literal
    restatement of the reference on the fuzz worlds: mixed materials, lights, hollow shells, a camera
    inside glass — same image within the parity bar."""
    from fuzz_worlds import fuzz_world_json
    from parity import pooled_atol
    sc = host.Scene.loads(fuzz_world_json(np.random.default_rng(1000 + kind), kind))
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    h_rgb, h_lin, h_st = hostsim.render(sc.ptr, mode=3 + 16)
    # (worlds with lights: the lane logic skips the light rays whose result raytracer.rs:124 discards — fewer
    # segments than the literal restatement, same radiance)
    assert 0 < int(h_st["segments"]) <= int(o_st["segments"]) and len(sc.lights()) == 2
    assert_parity(h_rgb, h_lin, o_rgb, o_lin, f"fuzz world {kind}", atol=pooled_atol(sc.c.samples_per_pixel), flip_frac=2e-3)


def test_more_than_65535_spheres_take_the_wide_tables(hostsim, oracle, abi, host):
    """The reference accepts any object count (a Vec<Sphere>, raytracer.rs:52-57).  Above 65 535 spheres the packed cell tables
    (u16 item lists) cannot name a sphere: build_grid switches to the WIDE format — 32-bit item lists, four words per cell —
    and the walk stays a walk (until round 5 such a scene fell back to the reference's full scan: 66 001 tests per segment).
    Same closest hits as brute force per segment (audit mode), same frame as the oracle."""
    n = 66000
    from fuzz_worlds import big_flat_world_json
    sc = host.Scene.loads(big_flat_world_json(n, np.random.default_rng(3)))
    assert sc.c.n_spheres == n + 1 > 65535
    info = (C.c_uint32 * 6)()
    assert hostsim.hostsim_grid_info(sc.ptr, info) == 0
    assert info[0] > 1 and info[2] > 1 and info[3] <= 8 and info[5] >= n     # a grid; only the ground is `large`; every small sphere listed
    assert hostsim.hostsim_grid_wide(sc.ptr) == 1
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    rgb, lin, st = hostsim.render(sc.ptr, None, 3)
    assert_parity(rgb, lin, o_rgb, o_lin, "66001 spheres")
    assert st["segments"] == o_st["segments"] and st["exact_tests"] < 30 * st["segments"]   # (the full scan: 66 001 per segment)
    _, _, st_audit = hostsim.render(sc.ptr, None, 4)
    assert st_audit["kernel_ms"] == 0.0, "grid walk and brute force disagree on some segment"


def test_the_every_ray_list_takes_big_spheres_only_when_a_radius_gap_follows(hostsim, abi, load_scene):
    """rt_tables.h build_grid: the `large` list (spheres every ray tests instead of finding them in the grid) takes the k <= 8
    biggest candidates, k the largest count after which the radii drop by half or more — the bulk left in the grid must be
    smaller for the promotion to buy anything.  The cover scene keeps ground + three r = 1 spheres; a world whose big spheres
    are a continuum keeps the ground only (until round 5: the biggest eight, eight tests per ray for nothing); outliers above
    a crowd of medium spheres are taken, the crowd is not."""
    info = (C.c_uint32 * 6)()
    sc = load_scene("cover", 60, 40, 1, 5)
    assert hostsim.hostsim_grid_info(sc.ptr, info) == 0 and info[3] == 4

    def world(radii):
        rng = np.random.default_rng(len(radii))
        spheres = (abi.RtSphere * (len(radii) + 1))()
        for i, r in enumerate(radii):
            spheres[i].center[:] = [float(rng.uniform(-40, 40)), float(r), float(rng.uniform(-40, 40))]
            spheres[i].radius = float(r)
            spheres[i].kind = abi.RT_MAT_LAMBERTIAN
        spheres[len(radii)].center[:] = [0.0, -1000.0, 0.0]
        spheres[len(radii)].radius = 1000.0
        scn = abi.RtScene(abi_version=abi.RT_ABI_VERSION, width=4, height=4, samples_per_pixel=1, max_depth=2, spheres=spheres, n_spheres=len(spheres))
        assert hostsim.hostsim_grid_info(C.byref(scn), info) == 0 and info[0] > 0
        return info[3]

    bulk = [0.2] * 400
    assert world(bulk + list(np.exp(np.random.default_rng(1).uniform(np.log(0.9), np.log(5.0), 60)))) == 1   # a continuum of big spheres: the ground only
    assert world(bulk + [3.0] * 40) == 1                                # forty equal big spheres: none of them
    assert world(bulk + [10.0] * 3 + [1.0] * 15) == 4                   # three outliers above a crowd of fifteen: ground + the three
    assert world(bulk + [1.0] * 3) == 4                                 # the cover scene's shape
    assert world(bulk + [1.0] * 9) == 1                                 # more candidates than the list holds and no gap inside them


def test_a_cell_with_more_items_than_the_packed_word_counts(hostsim, oracle, abi, host):
    """4 300 nearly coincident spheres share their cells: more than the 4 095 items a packed cell word can count.  Such a scene
    takes the wide tables too (until round 5: no grid, the full scan for every ray of the scene)."""
    from fuzz_worlds import crowded_cell_world_json
    sc = host.Scene.loads(crowded_cell_world_json())
    info = (C.c_uint32 * 6)()
    assert hostsim.hostsim_grid_info(sc.ptr, info) == 0 and info[0] > 1 and info[3] <= 8
    assert hostsim.hostsim_grid_wide(sc.ptr) == 1 and sc.c.n_spheres < 65535
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    rgb, lin, st = hostsim.render(sc.ptr, None, 3)
    assert_parity(rgb, lin, o_rgb, o_lin, "crowded cell")
    assert st["segments"] == o_st["segments"] and st["exact_tests"] < st["sphere_tests"]
    _, _, st_audit = hostsim.render(sc.ptr, None, 4)
    assert st_audit["kernel_ms"] == 0.0


@pytest.fixture
def wide_tables(monkeypatch):
    """RT_GRID_WIDE=1: the CPU build of the table builder (tests/hostsim, -DRT_TEST_PROBES) puts ANY world into the wide format"""
    monkeypatch.setenv("RT_GRID_WIDE", "1")


@pytest.mark.parametrize("kind", range(6))
def test_wide_tables_give_the_packed_tables_walk(hostsim, abi, host, kind, monkeypatch):
    """The wide format is only another encoding of the same grid: on the fuzz worlds (and the cover scene below) the walk
    visits the same cells and tests the same spheres — identical images, segment, test and step counts — and agrees with
    brute force on every segment."""
    from fuzz_worlds import fuzz_world_json
    sc = host.Scene.loads(fuzz_world_json(np.random.default_rng(2000 + kind), kind))
    assert hostsim.hostsim_grid_wide(sc.ptr) == 0
    p_rgb, p_lin, p_st = hostsim.render(sc.ptr, mode=3 + 16)
    monkeypatch.setenv("RT_GRID_WIDE", "1")
    assert hostsim.hostsim_grid_wide(sc.ptr) == 1
    w_rgb, w_lin, w_st = hostsim.render(sc.ptr, mode=3 + 16)
    assert np.array_equal(p_rgb, w_rgb) and np.array_equal(p_lin, w_lin)
    assert all(p_st[k] == w_st[k] for k in ("segments", "exact_tests", "grid_steps"))
    _, _, st_audit = hostsim.render(sc.ptr, mode=4)
    assert st_audit["kernel_ms"] == 0.0


def test_wide_tables_on_the_cover_scene(hostsim, load_scene, wide_tables):
    sc = load_scene("cover", 60, 40, 4, 50)
    assert hostsim.hostsim_grid_wide(sc.ptr) == 1
    _, _, st_audit = hostsim.render(sc.ptr, mode=4)
    assert st_audit["kernel_ms"] == 0.0 and st_audit["grid_steps"] > 0


TEXEL_RECORDS = [
    ("as shipped", {}),
    ("JSON width smaller than the decoded one", {"tex_w": 777}),
    ("JSON height larger than the decoded one: indices past the end are clamped and counted", {"tex_h": 5000}),
    ("negative offset: columns saturate at 0", {"h_offset": -0.3}),
    ("offset 3: columns beyond the row (texels of later rows; past the end only in the last rows)", {"h_offset": 3.0}),
    ("offset beyond the 4-byte-texel path's range: the u64 arithmetic on the RGB8 bytes", {"h_offset": 1500.0}),
    ("width beyond the 4-byte-texel path's range", {"tex_w": 1 << 25}),
    ("a one-pixel texture", {"tex_w": 1, "tex_h": 1}),
]


def texel_record_scene(load_scene, abi, changes, w=56, h=42, spp=2):
    """the reference's test scene (two Texture spheres, a sky texture) with the Texture records changed as given"""
    sc = load_scene("test", w, h, spp, 8)
    n_tex = 0
    for i in range(sc.c.n_spheres):
        s = sc.c.spheres[i]
        if s.kind == abi.RT_MAT_TEXTURE:
            n_tex += 1
            for k, v in changes.items():
                setattr(s, k, v)
    assert n_tex >= 1
    return sc


@pytest.mark.parametrize("what,changes", TEXEL_RECORDS, ids=[t[0].split(":")[0] for t in TEXEL_RECORDS])
def test_texel_paths_agree_with_the_oracle(what, changes, hostsim, oracle, abi, load_scene):
    """The device reads textures and sky as 4-byte texels with 32-bit index arithmetic (rt_core.h texels_fast /
    texture_albedo, sky_color) where the reference indexes RGB8 bytes in u64 (materials.rs:236-254, raytracer.rs:149-160);
    records outside that path's range keep the u64 form.  Both against the oracle's literal arithmetic, with Texture
    records that clamp, saturate and overflow the row: same pixels, same count of out-of-range fetches."""
    sc = texel_record_scene(load_scene, abi, changes)
    o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
    rgb, lin, st = hostsim.render(sc.ptr, None, 3)
    assert_parity(rgb, lin, o_rgb, o_lin, what)
    assert st["tex_oob"] == o_st["tex_oob"], what
    assert st["segments"] == o_st["segments"] - o_st["segments_discarded"]
    if changes.get("tex_h") == 5000:
        assert st["tex_oob"] > 0


def test_light_draw_words_and_the_open_high_word(hostsim, oracle, abi, host):
    """The light-sampling draw's 53 bits (raytracer.rs:100): a Glass hit takes words 2,3 of its slot-0 call, every other hit the
    HIGH word from word 3 of attempt 0's call (slot 1) and the low word from slot 0 word 2 (oracle/rt_oracle.c, "RNG addressing").
    tests/light_draw_cases.py holds seeds whose pixel-0 draw has the one high word in 2^32 that does not decide the comparison:
    (1) the fixture is what it says (through the oracle's own Philox), (2) the low word decides as recorded, (3) kernel logic
    (CPU build: both words always drawn) and oracle trace the same paths there.  The kernel's own decision from the high word, and
    its call for the low one, is the GPU test of the same name."""
    import ctypes as C
    import light_draw_cases as ldc
    L = oracle.lib(abi)
    assert ldc.OPEN_HIGH_WORD == 3865470566 and 0 < ldc.LOW_PART_BOUND < (1 << 21)
    for seed, samples in ldc.OPEN_SEEDS.items():
        u01, rg = (C.c_double * 2)(), (C.c_double * 3)()
        L.rt_oracle_draws(seed, 0, 0, 0, 1, u01, rg)
        assert int(u01[1] * 2.0 ** 32) == ldc.OPEN_HIGH_WORD                  # slot 1, word 3
        L.rt_oracle_draws(seed, 0, 0, 0, 0, u01, rg)
        low_part = int(u01[1] * 2.0 ** 53) & 0x1FFFFF                         # slot 0, word 2 >> 11
        assert (low_part >= ldc.LOW_PART_BOUND) == samples
        sc = host.Scene.loads(ldc.scene_json())
        sc.c.seed = seed
        o_rgb, o_lin, o_st = oracle.render(abi, sc.ptr)
        rgb, lin, st = hostsim.render(sc.ptr, None, 3)
        assert_parity(rgb, lin, o_rgb, o_lin, f"seed {seed}")
        assert st["segments"] == o_st["segments"] - o_st["segments_discarded"]
    assert len(set(ldc.OPEN_SEEDS.values())) == 2


def test_light_pool_index_arithmetic():
    """The device's light pools (rt_core.h, round 5) use full-rate integer arithmetic where a division would be: the index of
    the frame record at an LDS offset — (offset - first record) / 80 as ((x >> 4) * 13108) >> 16, exact because offsets are
    multiples of 16 and (y * ceil(2^16 / 5)) >> 16 == y // 5 for y < 2^14 — and the start word of a bitmap search —
    (16-bit hash * words) >> 16, which must land inside the bitmap for every word count a pool can have."""
    y = np.arange(0, 1 << 14, dtype=np.uint64)
    assert np.array_equal((y * 13108) >> 16, y // 5)
    for base_slots in (0, 32, 288, 1024):
        frames_off = 66592 + 256 + base_slots * 32          # any multiple of 16 does
        slots = np.arange(0, 1024, dtype=np.uint64)
        where = frames_off + slots * 80
        assert (where % 16 == 0).all() and where.max() < (1 << 18)
        assert np.array_equal((((where - frames_off) >> 4) * 13108) >> 16, slots)
    h = np.arange(0, 1 << 16, dtype=np.uint64)
    for words in range(1, 33):
        w = (h * words) >> 16
        assert w.min() == 0 and w.max() == words - 1
