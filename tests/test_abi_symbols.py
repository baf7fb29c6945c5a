"""The C-ABI libraries load on a CPU-only box and export every symbol include/rt_abi.h
declares; the product has no CPU compute path (calls without a GPU fail loudly)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SYMS = {"rt_scene_camera", "rt_scene_load_timings", "rt_scene_load_file", "rt_scene_load_string", "rt_scene_get", "rt_scene_get_mut", "rt_scene_free", "rt_scene_to_json",
             "rt_host_last_error", "rt_camera_derive", "rt_find_lights", "rt_jpeg_decode_file", "rt_jpeg_decode_mem", "rt_jpeg_last_error",
             "rt_png_write_rgb8", "rt_free"}


def declared_functions(header="rt_abi.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text))
    return names - {"rt_tiles_local_rows", "rt_tiles_global_row", "rt_tiles_stacked_row"}  # static inline helpers


def test_every_declared_symbol_is_exported(pkg):
    """include/rt_abi.h = the seam: every function it declares is exported by the two PRODUCT libraries; include/rt_abi_test.h =
    the lab (device probes, debug calls): exported by librt_hip_probe.so — which carries the whole product API too, a scene the
    debug calls look into is created through it — and by nothing a drop-in host links."""
    host = C.CDLL(os.path.join(ROOT, "rust-raytracer_amd", "librt_host.so"))
    hip = C.CDLL(pkg.hip.LIB_PATH)
    probe = C.CDLL(pkg.hip.PROBE_LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in sorted(names):
        lib = host if n in HOST_SYMS else hip
        assert hasattr(lib, n), f"{n} not exported"
        if n not in HOST_SYMS:
            assert hasattr(probe, n), f"{n} not exported by the probe library"
    lab = declared_functions("rt_abi_test.h")
    assert len(lab) == 7 and not (lab & names), sorted(lab)
    for n in sorted(lab):
        assert hasattr(probe, n), f"{n} not exported by librt_hip_probe.so"
        assert not hasattr(hip, n) and not hasattr(host, n), f"{n} is exported by a product library"
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "_probe" not in out and "rt_hip_debug" not in out, [l for l in out.splitlines() if "_probe" in l or "debug" in l]
    # and nothing the product exports is undeclared: every defined rt_* symbol of librt_hip.so is in the header
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("rt_")}
    assert exported <= names, sorted(exported - names)


def _rust_structs(text):
    """{name: [(field, rust type)]} of the #[repr(C)] structs in INTEGRATION.md's Rust shim"""
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in re.split(r",(?![^\[]*\])", m.group(2)):
            f = f.strip()
            if f:
                name, ty = f.split(":", 1)
                fields.append((name.strip().replace("pub ", ""), ty.strip()))
        out[m.group(1)] = fields
    return out


def _ctype_of(rust):
    prim = {"u8": C.c_uint8, "u32": C.c_uint32, "u64": C.c_uint64, "f32": C.c_float, "f64": C.c_double, "i32": C.c_int32}
    m = re.fullmatch(r"\[(\w+);\s*(\d+)\]", rust)
    if m:
        return prim[m.group(1)] * int(m.group(2))
    if rust.startswith("*const") or rust.startswith("*mut"):
        return C.c_void_p
    return prim[rust]


def test_rust_shim_in_integration_md_matches_the_header(abi):
    """The Rust binding cannot be compiled here (no rustc): machine-check its #[repr(C)] struct literals against
    abi.py (itself checked against the C header below) — field names, order, sizes, total size.  A stale
    `prof_cycles: [u64; 8]` (round 1) would make rt_hip_wait's memset write past the Rust struct."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rs = _rust_structs(text)
    assert set(rs) >= {"RtSphere", "RtTexture", "RtScene", "RtStats", "RtRowTiles"}, sorted(rs)
    for name, fields in rs.items():
        ct = getattr(abi, name)
        want = [(n, C.sizeof(t)) for n, t in ct._fields_]
        got = [(n, C.sizeof(_ctype_of(t))) for n, t in fields]
        assert got == want, (name, got, want)
        mirror = type("M" + name, (C.Structure,), {"_fields_": [(n, _ctype_of(t)) for n, t in fields]})
        assert C.sizeof(mirror) == C.sizeof(ct), name
    assert f"abi_version: {abi.RT_ABI_VERSION}" in text           # the version the shim writes into RtScene
    assert "rt_abi_sizeof" in text and "rt_abi_version" in text   # and it checks the library's layout at start-up


def test_abi_sizeof_export(pkg, abi):
    """rt_abi_sizeof / rt_abi_version: what a foreign binding compares its own struct sizes with"""
    L = pkg.hip.lib()
    for name in ("RtSphere", "RtTexture", "RtScene", "RtRowTiles", "RtStats", "RtGroupInfo", "RtGroupRank"):
        assert L.rt_abi_sizeof(name.encode()) == C.sizeof(getattr(abi, name))
    assert L.rt_abi_sizeof(b"NoSuchStruct") == 0 and L.rt_abi_version() == abi.RT_ABI_VERSION


def test_struct_layout_matches_header(abi, tmp_path):
    """ctypes mirrors == C sizeof/offsetof (compiled from the header with gcc)."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "rt_abi_test.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(RtSphere), sizeof(RtTexture),'
                   ' sizeof(RtScene), sizeof(RtRowTiles), sizeof(RtStats), offsetof(RtScene, spheres), offsetof(RtScene, seed), offsetof(RtSphere, albedo));'
                   'printf("%zu %zu %zu %zu %zu\\n", sizeof(RtGroupInfo), sizeof(RtGroupRank), offsetof(RtGroupInfo, device), offsetof(RtGroupRank, pci_bus_id), offsetof(RtGroupRank, kernel_ms));'
                   'printf("%zu %zu %zu %zu\\n", offsetof(RtScene, n_gpus), offsetof(RtStats, segments_discarded), offsetof(RtStats, gather_ms), offsetof(RtStats, prof_cycles));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(abi.RtSphere), C.sizeof(abi.RtTexture), C.sizeof(abi.RtScene), C.sizeof(abi.RtRowTiles), C.sizeof(abi.RtStats),
            abi.RtScene.spheres.offset, abi.RtScene.seed.offset, abi.RtSphere.albedo.offset,
            C.sizeof(abi.RtGroupInfo), C.sizeof(abi.RtGroupRank), abi.RtGroupInfo.device.offset, abi.RtGroupRank.pci_bus_id.offset, abi.RtGroupRank.kernel_ms.offset,
            abi.RtScene.n_gpus.offset, abi.RtStats.segments_discarded.offset, abi.RtStats.gather_ms.offset, abi.RtStats.prof_cycles.offset]
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
                     r'librt_oracle|libhostsim|rt_oracle_[a-z_0-9]+\s*\(')
    bad = []
    for base in ("rust-raytracer_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    t = open(os.path.join(dp, f), errors="replace").read()
                    if pat.search(t):
                        bad.append(os.path.join(dp, f))
                    if "dlopen" in t:  # the only library the product loads at run time is RCCL (multi-GPU gather)
                        assert f == "rt_hip_group.hip", f
                        libs = re.findall(r'"([^"]*\.so[^"]*)"', t)
                        assert libs and all("rccl" in x for x in libs), libs
    assert bad == [], bad
    # and the built product libraries carry no dependency on / symbol of the oracle
    for so in ("librt_hip.so", "librt_hip_probe.so", "librt_host.so"):
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


def test_group_deinterleave_layout_inverts_the_rank_packing(pkg, abi):
    """The de-interleave of rt_hip_group_* (frame[y] <- gather buffer row), as librt_hip.so compiled it, against the
    packing every rank uses (RtRowTiles{2, r, G}: abi.tiles_global_rows): every scanline is found exactly once, for
    ragged heights and more ranks than tiles.  Runs without a GPU (pure layout arithmetic of the library)."""
    L = pkg.hip.lib()
    tr = C.c_uint32()
    L.rt_hip_group_stacked_row(0, 1, 1, C.byref(tr))
    assert tr.value == 2
    for h in (1, 2, 3, 7, 45, 800, 2160):
        for world in (1, 2, 3, 8, 16):
            shards = [abi.tiles_global_rows(h, abi.RtRowTiles(tr.value, r, world)) for r in range(world)]
            pad = max(len(s) for s in shards)
            seen = {}
            for r, rows in enumerate(shards):
                for lr, y in enumerate(rows):
                    seen[r * pad + lr] = y                      # what rank r wrote at packed row lr
            got = [seen.get(L.rt_hip_group_stacked_row(y, world, pad, None)) for y in range(h)]
            assert got == list(range(h)), (h, world)


def test_build_info_describes_the_sources_in_the_tree(pkg):
    """BUILD_INFO.json (next to the libraries; what bench.py and the PMC tools quote) carries the hash of the kernel
    sources the in-tree librt_hip.so was compiled from — a stale library, or counters of another kernel, show up as a
    mismatch (`roofline.counters.same_kernel_sources` on the bench line) instead of silently describing different code."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rt_build", os.path.join(root, "rust-raytracer_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    h = b.kernel_src_hash()
    assert len(h) == 12 and h == b.kernel_src_hash()
    info_path = os.path.join(root, "rust-raytracer_amd", "BUILD_INFO.json")
    if os.path.exists(info_path):   # (written by build(); a snapshot without .git keeps the file that travelled with it)
        info = json.load(open(info_path))
        assert info.get("kernel_src_hash") == h, "librt_hip.so is older than the kernel sources: run build()"
