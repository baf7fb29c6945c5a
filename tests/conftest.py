import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _cwd_repo_root():
    # scene files reference textures relative to the cwd, like the reference's `data/...`
    os.chdir(ROOT)
    yield


@pytest.fixture(scope="session")
def pkg():
    p = graft.load_package()
    if not os.path.exists(os.path.join(graft.PKG_DIR, "librt_host.so")):
        graft.build()
    return p


@pytest.fixture(scope="session")
def abi(pkg):
    return pkg.abi


@pytest.fixture(scope="session")
def host(pkg):
    pkg.host.lib()
    return pkg.host


@pytest.fixture(scope="session")
def oracle(abi):
    o = graft.load_oracle()
    o.lib(abi)
    return o


@pytest.fixture(scope="session")
def hostsim(abi):
    """CPU build of the kernel's per-lane logic (development check; tests only)."""
    d = os.path.join(ROOT, "tests", "hostsim")
    so = os.path.join(d, "libhostsim.so")
    src = os.path.join(d, "hostsim.cpp")
    deps = [src] + [os.path.join(ROOT, "rust-raytracer_amd", "csrc", "hip", f) for f in ("rt_core.h", "rt_tables.h")]
    deps.append(os.path.join(ROOT, "include", "rt_abi.h"))
    deps.append(os.path.join(ROOT, "rust-raytracer_amd", "csrc", "common", "rt_atan2.h"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mfma", "-fopenmp", "-Wno-unknown-pragmas",
                        "-DRT_TEST_PROBES",   # (rt_tables.h: RT_GRID_WIDE=1 in the environment puts any world through the wide table format)
                        "-DRT_DEV_KNOBS",     # (RT_GRID_N: a forced grid shape — test_rays_from_far_away_walk_the_grid)
                        "-shared", src, "-o", so], check=True)
    L = C.CDLL(so)
    L.hostsim_render.argtypes = [C.POINTER(abi.RtScene), C.POINTER(abi.RtRowTiles), C.c_void_p, C.c_void_p,
                                 C.POINTER(abi.RtStats), C.c_int]
    L.hostsim_cull_disc.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(abi.RtSphere)]
    L.hostsim_cull_disc.restype = C.c_float
    L.hostsim_exact_root.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(abi.RtSphere), C.c_double, C.c_double]
    L.hostsim_exact_root.restype = C.c_double
    L.hostsim_hit_prefix.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(abi.RtSphere)]
    L.hostsim_hit_prefix.restype = C.c_int
    L.hostsim_texels.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.c_double, C.c_uint64, C.c_uint64, C.c_void_p]
    L.hostsim_texels.restype = None
    L.hostsim_div_by_recip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.hostsim_div255.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.hostsim_range_m1_1.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.hostsim_u01_53.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.hostsim_sample_to_fixed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.hostsim_grid_info.argtypes = [C.POINTER(abi.RtScene), C.POINTER(C.c_uint32)]
    L.hostsim_grid_mode.argtypes = [C.POINTER(abi.RtScene), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.hostsim_hit_world.argtypes = [C.POINTER(abi.RtScene), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]

    def render(scene_ptr, tiles=None, mode=1):
        sc = scene_ptr.contents
        rows = abi.tiles_local_rows(sc.height, tiles)
        rgb = np.zeros((rows, sc.width, 3), np.uint8)
        lin = np.zeros((rows, sc.width, 3), np.float32)
        st = abi.RtStats()
        rc = L.hostsim_render(scene_ptr, C.byref(tiles) if tiles is not None else None, rgb.ctypes.data, lin.ctypes.data,
                              C.byref(st), mode)
        assert rc == 0
        return rgb, lin, st.as_dict()

    L.render = render
    return L


SCENES = {
    "cover": "scenes/cfg2_cover_1200x800_spp128.json",
    "test": "scenes/cfg1_test_800x600_spp16.json",
    "cover4k_tex": "scenes/cfg3_cover_4k_textured.json",
}


@pytest.fixture(scope="session")
def load_scene(host):
    def _load(name, width=None, height=None, spp=None, depth=None, seed=None):
        sc = host.Scene.load(SCENES.get(name, name))
        if width:
            sc.c.width = width
        if height:
            sc.c.height = height
        if spp:
            sc.c.samples_per_pixel = spp
        if depth is not None:
            sc.c.max_depth = depth
        if seed is not None:
            sc.c.seed = seed
        return sc
    return _load


def dvec(*v):
    return (C.c_double * len(v))(*v)
