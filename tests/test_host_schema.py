"""Host plumbing (librt_host.so): the JSON scene schema round-trips of the reference
(config.rs:78-147, sphere.rs:91-137, camera.rs:125-141, materials.rs:279-283), error
behaviour of main.rs:14-15, JPEG decode and PNG output."""
import json
import os

import numpy as np
import pytest

CFG_DEFAULT_SKY = ('{"width":100,"height":100,"samples_per_pixel":1,"max_depth":1,"sky":{"texture":""},"camera":{"look_from":{"x":0.0,"y":0.0,"z":0.0},'
                   '"look_at":{"x":0.0,"y":0.0,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":90.0,"aspect":1.0},"objects":[{"center":{"x":0.0,"y":0.0,"z":-1.0},'
                   '"radius":0.5,"material":{"Lambertian":{"albedo":[0.8,0.3,0.3]}}}]}')
CFG_NULL_SKY = CFG_DEFAULT_SKY.replace('"sky":{"texture":""}', '"sky":null')


def test_config_roundtrip_default_sky(host, abi):
    """config.rs:78-102 test_to_json: exact serde_json string."""
    sc = host.Scene.loads(CFG_DEFAULT_SKY)
    assert sc.to_json() == CFG_DEFAULT_SKY
    assert sc.c.sky_mode == abi.RT_SKY_GRADIENT and sc.c.n_spheres == 1
    assert list(sc.c.spheres[0].albedo) == [np.float32(0.8), np.float32(0.3), np.float32(0.3)]


def test_config_roundtrip_null_sky_and_sky_texture(host, abi):
    """config.rs:104-147 test_sky_perms_to_from_json."""
    sc = host.Scene.loads(CFG_NULL_SKY)
    assert sc.to_json() == CFG_NULL_SKY and sc.c.sky_mode == abi.RT_SKY_NONE
    tex = CFG_DEFAULT_SKY.replace('"sky":{"texture":""}', '"sky":{"texture":"scenes/data/earth.jpg"}')
    sc = host.Scene.loads(tex)
    assert sc.c.sky_mode == abi.RT_SKY_TEXTURE and (sc.c.sky_w, sc.c.sky_h) == (2048, 1024)
    assert sc.to_json() == tex


def test_sphere_and_texture_json(host, abi):
    """sphere.rs:91-137: Lambertian sphere string; Texture serialises pixels as
    "/tmp/texture.jpg" (materials.rs:28-33) and loads them from the given path."""
    base = CFG_DEFAULT_SKY[:CFG_DEFAULT_SKY.index('"objects":')]
    lam = '{"center":{"x":0.0,"y":0.0,"z":0.0},"radius":1.0,"material":{"Lambertian":{"albedo":[0.5,0.5,0.5]}}}'
    sc = host.Scene.loads(base + '"objects":[' + lam + "]}")
    assert sc.to_json().endswith('"objects":[' + lam + "]}")
    tload = ('{"center":{"x":0.0,"y":0.0,"z":0.0},"radius":1.0,"material":{"Texture":{"albedo":[0.5,0.5,0.5],"pixels":"scenes/data/earth.jpg",'
             '"width":2048,"height":1024,"h_offset":0.0}}}')
    sc = host.Scene.loads(base + '"objects":[' + tload + "," + tload + "]}")
    assert sc.to_json().endswith(tload.replace("scenes/data/earth.jpg", "/tmp/texture.jpg") + "]}")
    assert sc.c.n_textures == 1  # same path decoded once, shared
    t = sc.c.textures[0]
    assert (t.width, t.height, t.nbytes) == (2048, 1024, 2048 * 1024 * 3)
    px = host.jpeg_decode("scenes/data/earth.jpg")
    assert np.array_equal(np.ctypeslib.as_array(t.rgb8, shape=(t.nbytes,)), px.reshape(-1))


def test_camera_and_metal_json(host):
    """camera.rs:125-141 (aspect `(800/600) as f64` == 1.0), materials.rs:279-283."""
    cam = '"camera":{"look_from":{"x":-4.0,"y":4.0,"z":1.0},"look_at":{"x":0.0,"y":0.0,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":160.0,"aspect":1.0}'
    metal = '{"center":{"x":0.0,"y":0.0,"z":0.0},"radius":1.0,"material":{"Metal":{"albedo":[0.8,0.8,0.8],"fuzz":2.0}}}'
    text = '{"width":8,"height":6,"samples_per_pixel":1,"max_depth":1,"sky":null,' + cam + ',"objects":[' + metal + "]}"
    sc = host.Scene.loads(text)
    assert sc.to_json() == text
    d = host.camera_derive([-4, 4, 1], [0, 0, -1], [0, 1, 0], 160.0, 1.0)
    assert list(sc.c.cam_origin) == d["origin"] and list(sc.c.cam_lower_left) == d["lower_left_corner"]
    assert list(sc.c.cam_horizontal) == d["horizontal"] and list(sc.c.cam_vertical) == d["vertical"]


def test_reference_scenes_load(host, abi):
    """config.rs:249-255 test_from_file + the committed benchmark configs."""
    sc = host.Scene.load("scenes/cfg1_test_800x600_spp16.json")
    c = sc.c
    assert (c.width, c.height, c.samples_per_pixel, c.max_depth) == (800, 600, 16, 8)
    assert c.n_spheres == 7 and c.n_textures == 2 and c.sky_mode == abi.RT_SKY_TEXTURE and (c.sky_w, c.sky_h) == (2410, 1205)
    kinds = [c.spheres[i].kind for i in range(7)]
    assert kinds == [abi.RT_MAT_TEXTURE, abi.RT_MAT_TEXTURE, abi.RT_MAT_METAL, abi.RT_MAT_LIGHT, abi.RT_MAT_METAL, abi.RT_MAT_GLASS, abi.RT_MAT_GLASS]
    assert c.spheres[6].radius == -0.45 and sc.lights() == [3]
    sc = host.Scene.load("scenes/cfg2_cover_1200x800_spp128.json")
    c = sc.c
    assert (c.width, c.height, c.samples_per_pixel, c.max_depth, c.n_spheres) == (1200, 800, 128, 50, 484)
    kinds = np.array([c.spheres[i].kind for i in range(484)])
    assert [(kinds == k).sum() for k in range(5)] == [407, 56, 21, 0, 0]  # SURVEY §2 census
    # a serde round trip of the whole file reproduces every number (shortest-repr floats)
    again = host.Scene.loads(sc.to_json())
    assert again.to_json() == sc.to_json()
    assert json.loads(sc.to_json()) == json.load(open("scenes/cfg2_cover_1200x800_spp128.json"))


@pytest.mark.parametrize("text,code", [
    ("{", "RT_ERR_PARSE"), ('{"width":1}', "RT_ERR_PARSE"), (CFG_DEFAULT_SKY.replace('"width":100', '"width":-1'), "RT_ERR_PARSE"),
    (CFG_DEFAULT_SKY.replace('"width":100', '"width":1.5'), "RT_ERR_PARSE"), (CFG_DEFAULT_SKY.replace("Lambertian", "Plastic"), "RT_ERR_PARSE"),
    (CFG_DEFAULT_SKY.replace('[0.8,0.3,0.3]', '[0.8,0.3]'), "RT_ERR_PARSE"),
    (CFG_DEFAULT_SKY.replace('"sky":{"texture":""}', '"sky":{"texture":"nope.jpg"}'), "RT_ERR_TEXTURE"),
    (CFG_DEFAULT_SKY.replace('"radius":0.5', '"radius":1e999'), "RT_ERR_PARSE"),   # serde_json: "number out of range"
    ("[" * 100000, "RT_ERR_PARSE"),                                               # serde_json: recursion limit
])
def test_errors_instead_of_panics(host, abi, text, code):
    """main.rs:14-15 / materials.rs:214 expect() panics become error codes."""
    with pytest.raises(host.RtError) as e:
        host.Scene.loads(text)
    assert e.value.code == getattr(abi, code)


def test_missing_file(host, abi):
    with pytest.raises(host.RtError) as e:
        host.Scene.load("/nonexistent/scene.json")
    assert e.value.code == abi.RT_ERR_IO and "Unable to read config file." in str(e.value)


def test_f32_field_beyond_f32_range_is_infinity_and_serialises_as_null(host):
    """serde_json reads the literal as f64 and casts (`as f32`): 1e39 is a valid f64 and becomes +inf in an albedo;
    to_string writes non-finite floats as null.  (Found by fuzzing under ASan: the number formatter used to index past
    the text "inf".)"""
    sc = host.Scene.loads(CFG_DEFAULT_SKY.replace("[0.8,0.3,0.3]", "[1e39,-1e39,0.3]"))
    a = sc.c.spheres[0].albedo
    assert a[0] == float("inf") and a[1] == float("-inf")
    assert '"albedo":[null,null,0.3]' in sc.to_json()


def test_unknown_fields_ignored_and_int_floats(host):
    """serde ignores unknown fields and accepts integer literals for f64 fields."""
    t = CFG_DEFAULT_SKY.replace('"radius":0.5', '"radius":2,"extra":[1,{"a":null}]')
    assert host.Scene.loads(t).c.spheres[0].radius == 2.0


@pytest.mark.parametrize("name", ["earth", "moon", "beach"])
def test_jpeg_decoder_close_to_libjpeg(host, name):
    """materials.rs:213-219: texel values are decoder-specific (unpinned by the reference);
    ours stay within a few levels of libjpeg (PIL) incl. 4:2:0 fancy upsampling (beach)."""
    from PIL import Image
    a = host.jpeg_decode(f"scenes/data/{name}.jpg").astype(int)
    b = np.asarray(Image.open(f"scenes/data/{name}.jpg").convert("RGB")).astype(int)
    assert a.shape == b.shape
    d = np.abs(a - b)
    assert d.max() <= 4 and d.mean() < 0.1


def test_png_write_roundtrip(host, tmp_path):
    """raytracer.rs:33-42 write_image: RGB8 PNG, lossless."""
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    host.png_write(p, img)
    back = Image.open(p)
    assert back.mode == "RGB" and back.size == (53, 37) and np.array_equal(np.asarray(back), img)


def _png_decode_by_hand(path):
    """stdlib-only decoder of 8-bit RGB non-interlaced PNGs: every chunk's CRC, consecutive IDAT chunks as ONE zlib stream
    (Adler-32 checked by zlib), the five filter types"""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr, kinds = 8, [], None, []
    while pos < len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        kind, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(kind + body) == crc, kind
        kinds.append(kind)
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        pos += 12 + n
    assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND" and set(kinds[1:-1]) == {b"IDAT"}   # IDAT chunks consecutive
    w, h, depth, colour, comp, filt, lace = ihdr
    assert (depth, colour, comp, filt, lace) == (8, 2, 0, 0, 0)
    raw = zlib.decompress(b"".join(idat))
    stride = w * 3
    assert len(raw) == (stride + 1) * h
    out = np.zeros((h, stride), np.uint8)
    for y in range(h):
        f, line = raw[y * (stride + 1)], np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(stride, np.int32)
        cur = np.zeros(stride, np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            assert f in (1, 3, 4)
            for i in range(stride):
                a = cur[i - 3] if i >= 3 else 0
                b, c = prev[i], (prev[i - 3] if i >= 3 else 0)
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
        out[y] = cur
    return out.reshape(h, w, 3), len(idat)


@pytest.mark.parametrize("deflate", [None, "rle", "default", "huffman"])
@pytest.mark.parametrize("threads", [None, "1", "3"])
def test_png_writer_bands_strategies_threads(host, tmp_path, monkeypatch, deflate, threads):
    """rt_png_write_rgb8 (raytracer.rs:33-42: what counts is the DECODED pixels): frames of several bands (one IDAT chunk each + the
    Adler-32's), every RT_PNG_DEFLATE strategy and thread count — decoded by PIL and by a stdlib-only decoder that checks every
    chunk CRC and the stream's Adler-32; the file does not depend on the thread count."""
    from PIL import Image
    if deflate:
        monkeypatch.setenv("RT_PNG_DEFLATE", deflate)
    if threads:
        monkeypatch.setenv("RT_PNG_THREADS", threads)
    rng = np.random.default_rng(1)
    y, x = np.mgrid[0:150, 0:421]
    smooth = np.stack([x * 255 // 421, y * 255 // 150, (x + y) * 255 // 571], -1).astype(np.int16)
    for name, img in (("noise_on_gradient", np.clip(smooth + rng.integers(-9, 10, smooth.shape), 0, 255).astype(np.uint8)),
                      ("flat", np.full((150, 421, 3), 7, np.uint8)), ("one_row", rng.integers(0, 256, (1, 5, 3), dtype=np.uint8)),
                      ("one_column", rng.integers(0, 256, (300, 1, 3), dtype=np.uint8))):
        p = str(tmp_path / f"{name}.png")
        host.png_write(p, img)
        assert np.array_equal(np.asarray(Image.open(p)), img), name
        back, n_idat = _png_decode_by_hand(p)
        assert np.array_equal(back, img), name
        if name == "noise_on_gradient":
            assert n_idat >= 3      # (several bands + the Adler-32's chunk)
            ref = tmp_path / "ref.png"
            monkeypatch.setenv("RT_PNG_THREADS", "2")
            host.png_write(str(ref), img)
            if threads:
                monkeypatch.setenv("RT_PNG_THREADS", threads)
            else:
                monkeypatch.delenv("RT_PNG_THREADS")
            assert open(p, "rb").read() == ref.read_bytes()   # band layout depends on the thread CAP only through "at most 8 bands per thread": not here


def test_png_writer_rejects_an_unknown_strategy(host, tmp_path, monkeypatch):
    monkeypatch.setenv("RT_PNG_DEFLATE", "zopfli")
    with pytest.raises(Exception):
        host.png_write(str(tmp_path / "x.png"), np.zeros((2, 2, 3), np.uint8))


def _baseline_jpeg(w=40, h=24, subsampling=2, restart=0):
    import io
    from PIL import Image
    rng = np.random.default_rng(5)
    img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    b = io.BytesIO()
    kw = dict(restart_marker_blocks=restart) if restart else {}
    img.save(b, "JPEG", quality=80, subsampling=subsampling, progressive=False, optimize=False, **kw)
    return b.getvalue()


def _segments(data):
    """[(marker, offset of 0xFF, segment length incl. the 2 length bytes)] up to SOS"""
    out, p = [], 2
    while p + 4 <= len(data):
        assert data[p] == 0xFF
        m, L = data[p + 1], (data[p + 2] << 8) | data[p + 3]
        out.append((m, p, L))
        if m == 0xDA:
            break
        p += 2 + L
    return out


def test_malformed_jpeg_is_rejected_not_overrun(host, abi):
    """a crafted texture must fail with RT_ERR_TEXTURE like jpeg-decoder's Err (materials.rs:214-217
    then panics on it), never write or read out of bounds: over-subscribed DHT code lengths, SOF / SOS / DRI segments
    shorter than the fields read from them, truncated files, and random byte flips."""
    good = _baseline_jpeg()
    assert host.jpeg_decode_mem(good).shape == (24, 40, 3)
    segs = _segments(good)
    kinds = {m for m, _, _ in segs}
    assert {0xC0, 0xC4, 0xDA, 0xDB} <= kinds

    def expect_fail(data, what):
        with pytest.raises(host.RtError) as e:
            host.jpeg_decode_mem(bytes(data))
        assert e.value.code == abi.RT_ERR_TEXTURE, what

    # (1) over-subscribed Huffman table: 255 codes of length 1 (the r1 decoder wrote ~64 KB past a 512-byte array)
    m, p, L = next(s for s in segs if s[0] == 0xC4)
    bad = bytearray(good)
    bad[p + 5] = 255                       # bits[1] of the first table
    expect_fail(bad, "over-subscribed DHT")
    bad = bytearray(good)
    bad[p + 5 + 1] = 5                     # 5 codes of length 2 (> 4 possible)
    expect_fail(bad, "over-subscribed DHT, length 2")
    # (2) segments declared shorter than the fields the parser reads from them
    for marker, short in ((0xC0, 4), (0xC0, 9), (0xDA, 2), (0xDA, 4)):
        m, p, L = next(s for s in segs if s[0] == marker)
        bad = bytearray(good[:p + 2]) + bytes([0, short]) + bytearray(good[p + 4:p + 2 + short]) + bytearray(good[p + 2 + L:])
        expect_fail(bad, f"short segment {marker:#x} L={short}")
    # a DRI segment with no payload, placed before SOS
    m, p, L = next(s for s in segs if s[0] == 0xDA)
    expect_fail(good[:p] + bytes([0xFF, 0xDD, 0, 2]) + good[p:], "empty DRI")
    # (2b) ADVICE r2 (medium): a frame header that claims 65535 x 65535 pixels in a tiny file must be refused BEFORE the
    # coefficient store is allocated (8 GiB per component, std::bad_alloc across the extern "C" boundary), and so must
    # any size the file could not possibly back with entropy-coded data
    m, p, L = next(s for s in segs if s[0] == 0xC0)
    for hh, ww in ((0xFFFF, 0xFFFF), (0x4000, 0x4000), (2048, 2048)):
        bad = bytearray(good)
        bad[p + 5:p + 9] = bytes([hh >> 8, hh & 255, ww >> 8, ww & 255])
        expect_fail(bad, f"claimed {ww}x{hh}")
    # (3) truncations: any prefix either fails cleanly or (entropy data cut short) decodes with zero-filled bits
    for cut in list(range(0, 64)) + list(range(64, len(good), 37)):
        try:
            host.jpeg_decode_mem(good[:cut])
        except host.RtError as e:
            assert e.code == abi.RT_ERR_TEXTURE
    # (4) random corruption of the headers and the scan, with and without restart markers / subsampling
    rng = np.random.default_rng(11)
    for base in (good, _baseline_jpeg(33, 17, 0), _baseline_jpeg(48, 32, 1, restart=2)):
        for _ in range(400):
            bad = bytearray(base)
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(2, len(bad)))] = int(rng.integers(0, 256))
            try:
                out = host.jpeg_decode_mem(bytes(bad))
                assert out.ndim == 3 and out.shape[2] == 3
            except host.RtError as e:
                assert e.code == abi.RT_ERR_TEXTURE


def test_progressive_and_multiscan_jpeg(host):
    """jpeg-decoder (what materials.rs:213-219 calls) reads progressive JPEGs; so does the C++ stand-in: SOF2 with
    spectral selection and successive approximation (DC / AC first and refinement scans, end-of-band runs, restart
    intervals), scans of one component walking that component's own block grid (4:4:4, 4:2:2, 4:2:0, greyscale, sizes
    that are not multiples of the MCU).  Texel values are decoder-specific; ours stay within a few levels of libjpeg."""
    import io
    from PIL import Image
    rng = np.random.default_rng(0)

    def img(w, h):
        a = rng.integers(0, 256, (h // 4 + 1, w // 4 + 1, 3), dtype=np.uint8)
        return Image.fromarray(a).resize((w, h), Image.BICUBIC)

    for w, h in ((64, 48), (37, 29), (200, 133)):
        im = img(w, h)
        for ss in (0, 1, 2):
            for rs in (0, 3):
                for gray in (False, True):
                    b = io.BytesIO()
                    kw = dict(restart_marker_blocks=rs) if rs else {}
                    (im.convert("L") if gray else im).save(b, "JPEG", quality=85, subsampling=ss, progressive=True, **kw)
                    data = b.getvalue()
                    assert b"\xff\xc2" in data                      # really a progressive frame
                    ours = host.jpeg_decode_mem(data).astype(int)
                    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
                    d = np.abs(ours - ref)
                    assert ours.shape == ref.shape and d.max() <= 4 and d.mean() < 0.2, (w, h, ss, rs, gray, d.max(), d.mean())


def test_malformed_progressive_jpeg_is_rejected_not_overrun(host, abi):
    """random corruption of progressive files (scan headers, Huffman data, refinement scans): RT_ERR_TEXTURE or an
    image, never a crash (the same harness ran 120 000 mutations under ASan + UBSan)"""
    import io
    from PIL import Image
    rng = np.random.default_rng(13)
    a = rng.integers(0, 256, (8, 12, 3), dtype=np.uint8)
    im = Image.fromarray(a).resize((44, 29), Image.BICUBIC)
    for ss, rs in ((2, 0), (0, 0), (1, 2)):
        b = io.BytesIO()
        im.save(b, "JPEG", quality=80, subsampling=ss, progressive=True, **(dict(restart_marker_blocks=rs) if rs else {}))
        base = b.getvalue()
        for it in range(500):
            bad = bytearray(base)
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(2, len(bad)))] = int(rng.integers(0, 256))
            if it % 7 == 0:
                bad = bad[: int(rng.integers(0, len(bad)))]
            try:
                out = host.jpeg_decode_mem(bytes(bad))
                assert out.ndim == 3 and out.shape[2] == 3
            except host.RtError as e:
                assert e.code == abi.RT_ERR_TEXTURE


def test_unsupported_jpeg_kinds_say_so(host, abi):
    """arithmetic-coded / lossless frames (which jpeg-decoder 0.2 does not decode either, or only in later versions)
    name their frame type in the error"""
    good = _baseline_jpeg()
    m, p, L = next(s for s in _segments(good) if s[0] == 0xC0)
    bad = bytearray(good)
    bad[p + 1] = 0xC9                                   # SOF9: extended sequential, arithmetic coding
    with pytest.raises(host.RtError) as e:
        host.jpeg_decode_mem(bytes(bad))
    assert e.value.code == abi.RT_ERR_TEXTURE and "SOF9" in str(e.value)


def test_scene_load_timings(host, abi):
    """rt_scene_load_timings: where a load went (read, JSON parse, longest JPEG decode, total) — what the CLI prints under
    RT_STATS=1 and bench.py's `cli` object quotes.  The test scene decodes three JPEGs concurrently beside the parse."""
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L = host.lib()
    L.rt_scene_load_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.rt_scene_load_timings.restype = None
    for path, wants_jpeg in ((os.path.join(root, "scenes", "cfg1_test_800x600_spp16.json"), True), (os.path.join(root, "scenes", "cfg2_cover_1200x800_spp128.json"), False)):
        cwd = os.getcwd()
        os.chdir(root)    # (texture paths resolve relative to the working directory, like the reference's)
        try:
            sc = host.Scene.load(path)
        finally:
            os.chdir(cwd)
        out = (C.c_double * 4)()
        L.rt_scene_load_timings(sc._h, out)
        read_ms, json_ms, jpeg_ms, total_ms = list(out)
        assert read_ms > 0 and json_ms > 0 and total_ms >= json_ms and total_ms >= jpeg_ms
        assert (jpeg_ms > 1.0) == wants_jpeg, (path, jpeg_ms)
        assert total_ms < 5000


# ---------------------------------------------------------------------------------------------------------------------------
# corner cases of serde-derive + serde_json the shipped scenes never touch.  What serde does:
#   * a derived struct deserialises from a MAP (unknown keys ignored; a known key twice = "duplicate field"; a missing key =
#     "missing field", an Option field = None) or from a SEQUENCE of exactly its fields in declaration order
#     (point3d.rs:10-15 `"center":[0,0,0]`, materials.rs:56-57 `"Light":[]`, camera.rs:29-36, sphere.rs:18-23, config.rs:66-75);
#   * an externally tagged enum (materials.rs:35-42) is a map with exactly ONE key;
#   * usize fields (config.rs:67-70) take any u64 — a frame side beyond u32 is a valid document this build cannot render:
#     RT_ERR_UNSUPPORTED, not a parse error; samples_per_pixel is u32 in the reference: beyond it serde errors too.
_P = '{"x":0.0,"y":0.0,"z":-1.0}'
_LAMB = '{"Lambertian":{"albedo":[0.8,0.3,0.3]}}'
_SERDE_CASES = [
    # ---- sequence form of structs: accepted
    ("point_as_sequence", CFG_DEFAULT_SKY.replace('"center":' + _P, '"center":[0.0,0.0,-1.0]'), "ok"),
    ("camera_points_as_sequences", CFG_DEFAULT_SKY.replace('"look_at":' + _P, '"look_at":[0,0,-1]').replace('"vup":{"x":0.0,"y":1.0,"z":0.0}', '"vup":[0,1,0]'), "ok"),
    ("camera_params_as_sequence", CFG_DEFAULT_SKY.replace('"camera":{"look_from":{"x":0.0,"y":0.0,"z":0.0},"look_at":' + _P + ',"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":90.0,"aspect":1.0}',
                                                          '"camera":[[0,0,0],[0,0,-1],[0,1,0],90.0,1.0]'), "ok"),
    ("sphere_as_sequence", CFG_DEFAULT_SKY.replace('{"center":' + _P + ',"radius":0.5,"material":' + _LAMB + '}', '[[0,0,-1],0.5,' + _LAMB + ']'), "ok"),
    ("lambertian_payload_as_sequence", CFG_DEFAULT_SKY.replace(_LAMB, '{"Lambertian":[[0.8,0.3,0.3]]}'), "ok"),
    ("metal_payload_as_sequence", CFG_DEFAULT_SKY.replace(_LAMB, '{"Metal":[[0.8,0.3,0.3],0.25]}'), "ok"),
    ("glass_payload_as_sequence", CFG_DEFAULT_SKY.replace(_LAMB, '{"Glass":[1.5]}'), "ok"),
    ("light_as_empty_sequence", CFG_DEFAULT_SKY.replace(_LAMB, '{"Light":[]}'), "ok"),
    ("light_as_empty_map", CFG_DEFAULT_SKY.replace(_LAMB, '{"Light":{}}'), "ok"),
    ("light_map_with_unknown_key", CFG_DEFAULT_SKY.replace(_LAMB, '{"Light":{"watts":60}}'), "ok"),
    ("texture_payload_as_sequence", CFG_DEFAULT_SKY.replace(_LAMB, '{"Texture":[[1,1,1],"scenes/data/earth.jpg",2048,1024,0.75]}'), "ok"),
    ("sky_as_sequence", CFG_DEFAULT_SKY.replace('"sky":{"texture":""}', '"sky":[""]'), "ok"),
    ("config_as_sequence", '[100,100,1,1,null,' + CFG_DEFAULT_SKY[CFG_DEFAULT_SKY.index('"camera":') + 9:CFG_DEFAULT_SKY.index(',"objects"')] + ',' +
     CFG_DEFAULT_SKY[CFG_DEFAULT_SKY.index('"objects":') + 10:-1] + ']', "ok"),
    ("sky_field_missing_is_none", CFG_DEFAULT_SKY.replace('"sky":{"texture":""},', ''), "ok"),
    # ---- sequence form with the wrong length: "invalid length"; other wrong types
    ("point_sequence_too_short", CFG_DEFAULT_SKY.replace('"center":' + _P, '"center":[0.0,0.0]'), "RT_ERR_PARSE"),
    ("point_sequence_too_long", CFG_DEFAULT_SKY.replace('"center":' + _P, '"center":[0.0,0.0,-1.0,2.0]'), "RT_ERR_PARSE"),
    ("light_sequence_not_empty", CFG_DEFAULT_SKY.replace(_LAMB, '{"Light":[1]}'), "RT_ERR_PARSE"),
    ("light_null", CFG_DEFAULT_SKY.replace(_LAMB, '{"Light":null}'), "RT_ERR_PARSE"),
    ("point_as_number", CFG_DEFAULT_SKY.replace('"center":' + _P, '"center":3'), "RT_ERR_PARSE"),
    ("metal_sequence_missing_fuzz", CFG_DEFAULT_SKY.replace(_LAMB, '{"Metal":[[0.8,0.3,0.3]]}'), "RT_ERR_PARSE"),
    # ---- duplicate keys: "duplicate field" for a known key, fine for an ignored one
    ("duplicate_width", CFG_DEFAULT_SKY.replace('"width":100', '"width":8,"width":9'), "RT_ERR_PARSE"),
    ("duplicate_sky", CFG_DEFAULT_SKY.replace('"sky":{"texture":""}', '"sky":null,"sky":null'), "RT_ERR_PARSE"),
    ("duplicate_point_coordinate", CFG_DEFAULT_SKY.replace('"center":' + _P, '"center":{"x":0.0,"x":1.0,"y":0.0,"z":-1.0}'), "RT_ERR_PARSE"),
    ("duplicate_radius", CFG_DEFAULT_SKY.replace('"radius":0.5', '"radius":0.5,"radius":0.5'), "RT_ERR_PARSE"),
    ("duplicate_albedo", CFG_DEFAULT_SKY.replace('{"albedo":[0.8,0.3,0.3]}', '{"albedo":[0.8,0.3,0.3],"albedo":[0.1,0.1,0.1]}'), "RT_ERR_PARSE"),
    ("duplicate_enum_tag", CFG_DEFAULT_SKY.replace(_LAMB, '{"Light":{},"Light":{}}'), "RT_ERR_PARSE"),
    ("two_enum_tags", CFG_DEFAULT_SKY.replace(_LAMB, '{"Light":{},"Glass":{"index_of_refraction":1.5}}'), "RT_ERR_PARSE"),
    ("duplicate_unknown_key_is_ignored_twice", CFG_DEFAULT_SKY.replace('"radius":0.5', '"radius":0.5,"note":1,"note":2'), "ok"),
    # ---- usize / u32 ranges
    ("width_2_pow_32", CFG_DEFAULT_SKY.replace('"width":100', '"width":4294967296'), "RT_ERR_UNSUPPORTED"),
    ("height_u64_max", CFG_DEFAULT_SKY.replace('"height":100', '"height":18446744073709551615'), "RT_ERR_UNSUPPORTED"),
    ("width_beyond_u64", CFG_DEFAULT_SKY.replace('"width":100', '"width":18446744073709551616'), "RT_ERR_PARSE"),
    ("max_depth_2_pow_40", CFG_DEFAULT_SKY.replace('"max_depth":1', '"max_depth":1099511627776'), "RT_ERR_UNSUPPORTED"),
    ("spp_beyond_u32", CFG_DEFAULT_SKY.replace('"samples_per_pixel":1', '"samples_per_pixel":4294967296'), "RT_ERR_PARSE"),
    ("width_u32_max_is_a_config", CFG_DEFAULT_SKY.replace('"width":100', '"width":4294967295'), "ok"),
]


@pytest.mark.parametrize("name,text,want", _SERDE_CASES, ids=[c[0] for c in _SERDE_CASES])
def test_serde_corner_cases(host, abi, name, text, want):
    assert text != CFG_DEFAULT_SKY, name     # (the case's replacement took)
    if want == "ok":
        sc = host.Scene.loads(text)
        assert sc.c.n_spheres == 1
        ref = host.Scene.loads(CFG_DEFAULT_SKY)
        # the sequence forms describe the SAME scene as the map forms (where the case did not change a value)
        if name in ("point_as_sequence", "camera_points_as_sequences", "camera_params_as_sequence", "sphere_as_sequence", "lambertian_payload_as_sequence", "sky_as_sequence"):
            assert sc.to_json() == ref.to_json()
        if name == "config_as_sequence":
            assert sc.to_json() == host.Scene.loads(CFG_NULL_SKY).to_json()
        if name == "sky_field_missing_is_none":
            assert sc.c.sky_mode == abi.RT_SKY_NONE
        if name == "metal_payload_as_sequence":
            assert sc.c.spheres[0].kind == abi.RT_MAT_METAL and sc.c.spheres[0].fuzz_or_ior == 0.25
        if name == "texture_payload_as_sequence":
            s0 = sc.c.spheres[0]
            assert s0.kind == abi.RT_MAT_TEXTURE and (s0.tex_w, s0.tex_h, s0.h_offset) == (2048, 1024, 0.75) and sc.c.n_textures == 1
        if name.startswith("light_"):
            assert sc.c.spheres[0].kind == abi.RT_MAT_LIGHT and sc.lights() == [0]
    else:
        with pytest.raises(host.RtError) as e:
            host.Scene.loads(text)
        assert e.value.code == getattr(abi, want), (name, str(e.value))
        if name.startswith("duplicate_") and "enum" not in name:
            assert "duplicate field" in str(e.value)
        if "too_" in name or name in ("light_sequence_not_empty", "metal_sequence_missing_fuzz"):
            assert "invalid length" in str(e.value)
