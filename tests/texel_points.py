"""Hit points on a textured sphere for the texel tests (CPU: tests/test_core_cpu.py through tests/hostsim; GPU:
tests/test_gpu_parity.py through rt_hip_texel_probe): uniformly distributed directions plus points AIMED at what could
make the fast (u, v) of rt_core.h::texel_fast pick another texel than the exact path — column boundaries incl. the wrap
rot = 1, row boundaries, the poles, and the seams |x| = |z|, x = 0, z = 0 of the atan reduction."""
import numpy as np

# (tex_w, tex_h, h_offset, radius, centre): earth / moon as the shipped scenes use them, a hollow shell (negative radius), a
# huge and a tiny sphere far from the origin, odd sizes
CASES = ((2048, 1024, 0.75, 0.5, (0.0, 0.0, -1.0)), (2048, 1024, 0.75, 1.0, (4.0, 1.0, 0.0)),
         (4096, 2048, 0.0, -0.45, (-1.2, 0.0, -1.0)), (7, 3, 0.3, 100.0, (0.0, -100.5, -1.0)),
         (1, 1, 0.999, 1e-3, (1e3, 2e3, -5e2)), (100003, 50021, 0.5, 2.0, (0.0, 0.0, 0.0)))


def points(rng, n, w, h, h_off, radius, centre):
    """-> (points [n,3] f64, index from which the directions are un-aimed / uniform)"""
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    # aimed points: u on (or within a few ulps .. 1e-9 of) a column boundary, incl. the wrap; v on a row boundary
    k = rng.integers(0, w + 1, n // 4)
    u_t = (k / w - h_off) % 1.0 + rng.choice([0.0, 1e-16, -1e-16, 1e-13, -1e-13, 1e-11, -1e-11, 1e-9, -1e-9], n // 4)
    ang = (u_t - 0.5) * 2.0 * np.pi
    yy = rng.uniform(-0.999, 0.999, n // 4)
    j = rng.integers(0, max(1, h), n // 8)
    yy[: n // 8] = np.clip(1.0 - 2.0 * j / max(1, h - 1) + rng.choice([0.0, 1e-16, -1e-15, 1e-12, -1e-10], n // 8), -1, 1)
    rr = np.sqrt(np.maximum(0.0, 1.0 - yy * yy))
    d[: n // 4] = np.stack([rr * np.sin(ang), yy, rr * np.cos(ang)], axis=1)
    # poles and the seams |x| = |z|, x = 0, z = 0
    m = n // 4
    d[m:m + 1000] = [0.0, 1.0, 0.0]
    d[m:m + 1000, 0] = rng.normal(size=1000) * 1e-9
    d[m + 1000:m + 2000, 0] = d[m + 1000:m + 2000, 2] * rng.choice([1.0, -1.0], 1000)
    d[m + 2000:m + 3000, 0] = 0.0
    d[m + 3000:m + 4000, 2] = 0.0
    pts = np.ascontiguousarray(np.asarray(centre)[None, :] + abs(radius) * d * (1.0 + rng.normal(size=(n, 1)) * 1e-9))
    return pts, m + 4000
