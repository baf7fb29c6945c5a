"""The C-ABI libraries load on a CPU-only box and export every symbol include/rt_abi.h
declares; the product has no CPU compute path (calls without a GPU fail loudly)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SYMS = {"rt_scene_camera", "rt_scene_load_file", "rt_scene_load_string", "rt_scene_get", "rt_scene_get_mut", "rt_scene_free", "rt_scene_to_json",
             "rt_host_last_error", "rt_camera_derive", "rt_find_lights", "rt_jpeg_decode_file", "rt_jpeg_decode_mem",
             "rt_png_write_rgb8", "rt_free"}


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text))
    return names - {"rt_tiles_local_rows", "rt_tiles_global_row"}  # static inline helpers


def test_every_declared_symbol_is_exported(pkg):
    host = C.CDLL(os.path.join(ROOT, "rust-raytracer_amd", "librt_host.so"))
    hip = C.CDLL(pkg.hip.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in sorted(names):
        lib = host if n in HOST_SYMS else hip
        assert hasattr(lib, n), f"{n} not exported"


def test_struct_layout_matches_header(abi, tmp_path):
    """ctypes mirrors == C sizeof/offsetof (compiled from the header with gcc)."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "rt_abi.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(RtSphere), sizeof(RtTexture),'
                   ' sizeof(RtScene), sizeof(RtRowTiles), sizeof(RtStats), offsetof(RtScene, spheres), offsetof(RtScene, seed), offsetof(RtSphere, albedo));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(abi.RtSphere), C.sizeof(abi.RtTexture), C.sizeof(abi.RtScene), C.sizeof(abi.RtRowTiles), C.sizeof(abi.RtStats),
            abi.RtScene.spheres.offset, abi.RtScene.seed.offset, abi.RtSphere.albedo.offset]
    assert got == want


def test_no_cpu_fallback_without_gpu(pkg, load_scene):
    """On a box without a GPU the hot path must refuse, not quietly compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert pkg.hip.device_count() == 0
    sc = load_scene("cover", 16, 16, 1)
    with pytest.raises(pkg.host.RtError) as e:
        pkg.hip.HipScene(sc.ptr, 0)
    assert e.value.code == pkg.abi.RT_ERR_NO_DEVICE
    with pytest.raises(pkg.host.RtError):
        pkg.hip.render_rgb8(sc.ptr)


def test_product_does_not_reference_oracle():
    """oracle/ and tests/hostsim are test infrastructure: no product source may include,
    import, dlopen or link them (comments may mention the oracle as the parity checker)."""
    pat = re.compile(r'#\s*include\s*[<"][^>"]*(oracle|hostsim)|import\s+[\w.]*oracle|from\s+[\w.]*oracle|load_oracle\s*\(|'
                     r'librt_oracle|libhostsim|rt_oracle_[a-z_0-9]+\s*\(|dlopen')
    bad = []
    for base in ("rust-raytracer_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    t = open(os.path.join(dp, f), errors="replace").read()
                    if pat.search(t):
                        bad.append(os.path.join(dp, f))
    assert bad == [], bad
    # and the built product libraries carry no dependency on / symbol of the oracle
    for so in ("librt_hip.so", "librt_host.so"):
        out = subprocess.run(["nm", "-D", os.path.join(ROOT, "rust-raytracer_amd", so)], capture_output=True, text=True).stdout
        assert "rt_oracle" not in out and "hostsim" not in out


def test_tiles_helpers(abi):
    for h in (1, 7, 8, 9, 60, 800):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                t = abi.RtRowTiles(8, r, world) if world > 1 else None
                g = abi.tiles_global_rows(h, t)
                assert len(g) == abi.tiles_local_rows(h, t)
                rows += g
            assert sorted(rows) == list(range(h))
