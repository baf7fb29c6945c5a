"""The host-side code the drop-in keeps on the CPU — the JSON scene schema, the JPEG decoder, the table / grid builder and the
CPU build of the per-lane path logic (tests/hostsim) — under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY §5 "race
detection / sanitizers": the reference is safe Rust; this host side is C++).  The harnesses of tools/fuzz/ are built with
-fsanitize=address,undefined and run for a bounded number of mutations: malformed JPEGs (baseline + progressive), malformed
scene JSON, adversarial sphere sets through tables -> grid -> walk == brute force.  Any sanitizer report fails the run
(-fno-sanitize-recover, halt_on_error).  Longer runs: tools/fuzz/README.md."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Iinclude", "-w"]
ENV = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")


def _build(tmp_path, name, sources, extra=()):
    exe = str(tmp_path / name)
    r = subprocess.run(["g++", *SAN, *extra, *sources, "-o", exe], cwd=ROOT, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in r.stderr.lower() and "cannot find" in r.stderr.lower():
        pytest.skip("no sanitizer runtime in this image")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(cmd, timeout, env=None):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env or ENV)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout[-500:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_jpeg_decoder_under_sanitizers(tmp_path):
    exe = _build(tmp_path, "fuzz_jpeg", ["tools/fuzz/fuzz_jpeg.cpp", "rust-raytracer_amd/csrc/host/jpeg.cpp"])
    out = _run([exe, "11", "40", "scenes/data/earth.jpg", "scenes/data/beach.jpg", "scenes/data/moon.jpg"], 300)
    assert "ok" in out and "err" in out


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_scene_schema_under_sanitizers(tmp_path):
    exe = _build(tmp_path, "fuzz_scene", ["tools/fuzz/fuzz_scene.cpp", "rust-raytracer_amd/csrc/host/scene.cpp", "rust-raytracer_amd/csrc/host/jpeg.cpp"],
                 extra=("-fopenmp", "-lz", "-lpthread"))
    out = _run([exe, "11", "2500", "scenes/cfg2_cover_1200x800_spp128.json"], 300)     # (no textures: a mutation costs a parse, not three JPEG decodes)
    out2 = _run([exe, "12", "12", "scenes/cfg1_test_800x600_spp16.json"], 300)
    assert "ok" in out and "ok" in out2


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_tables_grid_and_walk_under_sanitizers(tmp_path):
    exe = _build(tmp_path, "fuzz_tables", ["tools/fuzz/fuzz_tables.cpp", "tests/hostsim/hostsim.cpp"], extra=("-ffp-contract=off", "-fopenmp", "-lpthread", "-DRT_TEST_PROBES"))
    out = _run([exe, "11", "10"], 600)
    assert out.strip()
    out = _run([exe, "12", "6"], 600, env=dict(ENV, RT_GRID_WIDE="1"))   # the same sets through the wide table format (32-bit item lists)
    assert out.strip()


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_png_writer_under_sanitizers(tmp_path):
    """rt_png_write_rgb8 (round 6: parallel bands, one IDAT chunk per band, a z_stream per thread): 81 frames around the band / thread
    boundaries x deflate strategies x thread caps under ASan + UBSan; every file inflated again and compared with its scanlines."""
    exe = _build(tmp_path, "fuzz_png", ["tools/fuzz/fuzz_png.cpp", "rust-raytracer_amd/csrc/host/scene.cpp", "rust-raytracer_amd/csrc/host/jpeg.cpp"],
                 extra=("-fopenmp", "-lz", "-lpthread"))
    out = _run([exe, "5", str(tmp_path / "f.png")], 600)
    assert "ok 81 files" in out
