"""Which way is UP and which way is EAST on a textured sphere — pinned against the reference's own binary.

raytracer/output/out.png is the only image the Rust renderer left of its Texture path (earth, moon in glass, mirror ground).
It cannot be reproduced pixel for pixel (other camera, other h_offset, unseeded RNG, rendered before light sampling existed:
SURVEY.md §8c) — but the pole at the top and the continents un-mirrored are independent of all that, and are exactly what
`v = n.y * 0.5 + 0.5` with the row flip `(1 - v) * (height - 1)` (sphere.rs:42, materials.rs:243) and
`u = atan2(n.x, n.z) / 2pi + 0.5` (sphere.rs:39-41) decide.  tests/golden/out_png_earth.npz (made by
tests/golden/make_out_png_stats.py where /root/reference exists) holds the earth's disc of that image on a view-space grid.
Here the EARTH OBJECT of the committed test scene (data/test_scene.json:37-47: earth.jpg, 2048 x 1024, h_offset 0.75) is
rendered alone under the gradient sky from sixteen directions, a 64 x 32 longitude-latitude map is assembled from the
renders, and the reference's disc — turned into the same kind of map for a few assumed camera elevations — is slid over it
in longitude.  What is correlated is a SHADING-FREE feature, land against ocean: (R - B) / (R + G + B) with every
latitude row's mean taken out (the zonal mean — ice caps, the dark Southern Ocean — is the same mirrored or not; the
east-west arrangement of the continents is not).  The best match must be good (>= 0.8) un-mirrored and poor (< 0.5)
mirrored, and the northern polar cap must be white in both while the equatorial belt is not.  The CPU test renders with the oracle, the GPU test with the kernel."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NLON, NLAT = 64, 32
SIZE, DIST, VFOV = 160, 6.0, 11.0     # a narrow view from far away: close to orthographic (the sphere has radius 0.5)


def _earth_object():
    cfg = json.load(open(os.path.join(ROOT, "scenes", "cfg1_test_800x600_spp16.json")))
    earth = [o for o in cfg["objects"] if "Texture" in o["material"] and o["material"]["Texture"]["pixels"].endswith("earth.jpg")]
    assert len(earth) == 1 and earth[0]["material"]["Texture"]["h_offset"] == 0.75
    return earth[0]


def _view_json(earth, az_deg, elev_deg, spp):
    c = earth["center"]
    az, el = np.radians(az_deg), np.radians(elev_deg)
    look_from = {"x": c["x"] + DIST * np.cos(el) * np.sin(az), "y": c["y"] + DIST * np.sin(el), "z": c["z"] + DIST * np.cos(el) * np.cos(az)}
    return json.dumps({"width": SIZE, "height": SIZE, "samples_per_pixel": spp, "max_depth": 4, "sky": {"texture": ""},
                       "camera": {"look_from": look_from, "look_at": c, "vup": {"x": 0.0, "y": 1.0, "z": 0.0}, "vfov": VFOV, "aspect": 1.0},
                       "objects": [earth]})


def _land_white(rgb):
    """(land-against-ocean chroma (R - B) / (R + G + B), whiteness min / max) of an RGB array — both free of shading"""
    a = rgb.astype(np.float64)
    return (a[..., 0] - a[..., 2]) / np.maximum(a.sum(axis=-1), 1.0), a.min(axis=-1) / np.maximum(a.max(axis=-1), 1.0)


def _accumulate(maps, lon, lat, lum, white, weight):
    i = np.floor((lon + np.pi) / (2 * np.pi) * NLON).astype(int) % NLON
    j = np.clip(np.floor((np.pi / 2 - lat) / np.pi * NLAT).astype(int), 0, NLAT - 1)     # row 0 = north pole
    for arr, val in ((maps[0], lum * weight), (maps[1], white * weight), (maps[2], weight)):
        np.add.at(arr, (j, i), val)


def our_map(render):
    """NLON x NLAT land chroma / whiteness of OUR earth: render(json_text) -> rgb8 [SIZE, SIZE, 3]"""
    earth = _earth_object()
    r_px = SIZE / 2.0 * (0.5 / DIST) / np.tan(np.radians(VFOV) / 2.0) / np.sqrt(1.0 - (0.5 / DIST) ** 2)   # silhouette radius in pixels
    maps = [np.zeros((NLAT, NLON)) for _ in range(3)]
    ys, xs = np.mgrid[0:SIZE, 0:SIZE]
    vx = (xs + 0.5 - SIZE / 2.0) / r_px
    vy = -(ys + 0.5 - SIZE / 2.0) / r_px
    inside = vx * vx + vy * vy < 0.8 ** 2          # the central part of the disc: least foreshortened
    vz = np.sqrt(np.maximum(0.0, 1.0 - vx * vx - vy * vy))
    views = [(az, 0) for az in range(0, 360, 45)] + [(az, el) for el in (60, -60) for az in range(0, 360, 90)]   # the equator belt, then the caps
    for az_deg, el_deg in views:
        rgb = render(_view_json(earth, az_deg, el_deg, 64))
        lum, white = _land_white(rgb)
        az, el = np.radians(az_deg), np.radians(el_deg)
        # the camera sits in direction c = (cos el sin az, sin el, cos el cos az) of the centre and looks at it (vup = +y):
        # right = (cos az, 0, -sin az), up = (-sin el sin az, cos el, -sin el cos az); a view-space normal is vx right + vy up + vz c
        nx = vx * np.cos(az) - vy * np.sin(el) * np.sin(az) + vz * np.cos(el) * np.sin(az)
        ny = vy * np.cos(el) + vz * np.sin(el)
        nz = -vx * np.sin(az) - vy * np.sin(el) * np.cos(az) + vz * np.cos(el) * np.cos(az)
        _accumulate(maps, np.arctan2(nx, nz)[inside], np.arcsin(np.clip(ny, -1, 1))[inside], lum[inside], white[inside], vz[inside])
    assert (maps[2] > 0).all()
    return maps[0] / maps[2], maps[1] / maps[2]


def reference_maps(elev_deg, mirrored):
    """the reference's disc as a (partial) lon-lat map, longitudes relative to its unknown view azimuth, for an assumed camera
    elevation; mirrored: east and west exchanged (what a wrong-handed u would look like)"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "out_png_earth.npz"))
    rgb = z["rgb"].astype(np.float64)
    lum, white = _land_white(np.nan_to_num(rgb))
    lum[~np.isfinite(rgb[..., 0])] = np.nan
    G = lum.shape[0]
    jj, ii = np.mgrid[0:G, 0:G]
    vx = (ii + 0.5) / G * 2.0 - 1.0
    vy = 1.0 - (jj + 0.5) / G * 2.0
    ok = np.isfinite(lum) & (vx * vx + vy * vy < 0.85 ** 2)
    vz = np.sqrt(np.maximum(0.0, 1.0 - vx * vx - vy * vy))
    e = np.radians(elev_deg)     # camera above the equator plane, looking down by e: view -> world is a rotation about x
    wy = vy * np.cos(e) + vz * np.sin(e)
    wz = -vy * np.sin(e) + vz * np.cos(e)
    wx = -vx if mirrored else vx
    maps = [np.zeros((NLAT, NLON)) for _ in range(3)]
    _accumulate(maps, np.arctan2(wx, wz)[ok], np.arcsin(np.clip(wy, -1, 1))[ok], lum[ok], white[ok], vz[ok])
    return maps


def best_match(ours_lum, mirrored):
    best = (-2.0, None, None)
    for elev in range(0, 50, 5):
        rl, _, rw = reference_maps(elev, mirrored)
        cover = rw > 0
        cover &= cover.sum(axis=1, keepdims=True) >= 6          # (rows with a handful of cells have no zonal mean worth removing)
        ref = np.where(cover, rl / np.maximum(rw, 1e-12), 0.0)
        w = np.where(cover, rw, 0.0)

        def anomalies(m):   # every latitude row minus its (weighted) mean over the covered cells
            return np.where(cover, m - (m * w).sum(axis=1, keepdims=True) / np.maximum(w.sum(axis=1, keepdims=True), 1e-12), 0.0)
        a = anomalies(ref)
        for shift in range(NLON):
            b = anomalies(np.roll(ours_lum, shift, axis=1))
            c = float(np.sum(w * a * b) / np.sqrt(np.sum(w * a * a) * np.sum(w * b * b)))
            if c > best[0]:
                best = (c, elev, shift)
    return best


def check_orientation(render, who):
    ours_lum, ours_white = our_map(render)
    plain = best_match(ours_lum, mirrored=False)
    mirror = best_match(ours_lum, mirrored=True)
    print(f"{who} vs the reference's out.png: best correlation {plain[0]:.3f} (elevation {plain[1]} deg, shift {plain[2]} / {NLON}); east-west mirrored {mirror[0]:.3f}")
    assert plain[0] >= 0.8 and mirror[0] < 0.5, (plain, mirror)
    # the pole: north of 73 deg N the map is ice — white — in OUR render as in the reference's (at its best elevation), the
    # belt between 34 N and the equator is blue ocean and coloured land; upside down (a missing v flip) the top rows of the
    # reference's view would show the open Southern Ocean instead
    rl, rwh, rw = reference_maps(plain[1], False)
    top, mid = slice(0, 3), slice(10, 16)
    ref_top_white = rwh[top].sum() / rw[top].sum()
    ref_mid_white = rwh[mid].sum() / rw[mid].sum()
    our_top_white, our_mid_white = ours_white[top].mean(), ours_white[mid].mean()
    print(f"  whiteness north of 73 N / 0-34 N: ours {our_top_white:.2f} / {our_mid_white:.2f}, reference {ref_top_white:.2f} / {ref_mid_white:.2f}")
    assert rw[top].sum() > 0 and ref_top_white > ref_mid_white + 0.1 and our_top_white > our_mid_white + 0.1


def test_oracle_texture_orientation_matches_the_reference_render(oracle, abi, host):
    def render(text):
        sc = host.Scene.loads(text)
        rgb, _, _ = oracle.render(abi, sc.ptr, want_linear=False)
        return rgb
    check_orientation(render, "CPU oracle")


@pytest.mark.gpu
def test_gpu_texture_orientation_matches_the_reference_render(pkg, host):
    import torch

    def render(text):
        sc = host.Scene.loads(text)
        gs = pkg.hip.HipScene(sc.ptr, 0)
        rgb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
        gs.render(rgb.data_ptr(), 0, None, torch.cuda.current_stream().cuda_stream)
        gs.wait()
        gs.close()
        return rgb.cpu().numpy()
    check_orientation(render, "GPU kernel")
