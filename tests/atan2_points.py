"""Arguments for the pin of the shared atan2 (rust-raytracer_amd/csrc/common/rt_atan2.h): ~1.1 M (y, x) pairs built from
a splitmix64 stream and IEEE operations only (+ - * / sqrt, bit casts) — no libm call, no numpy random generator — so the
same bits come out on every machine and numpy version.  tests/golden/make_atan2_fixture.py computes the correctly rounded
results of these pairs with mpmath; tests/test_oracle_kat.py and the GPU probe check the routine against them.

Categories (sphere.rs:35-43 calls atan2(n.x, n.z) of a unit vector; the routine reduces t = min/max, k = round(32 t),
z = (t - k/32) / (1 + t k/32), then the quadrant):
  unit     components of random unit vectors (what the kernel passes), all octants
  plain    both arguments uniform in [-1, 1)
  wide     magnitudes spread over 2^-300 .. 2^300
  knots    min/max = k/32 exactly (k = 0..32) and a few ulps beside it, every octant and swap
  halfway  min/max at (k + 1/2)/32, where k = round(32 t) changes (|z| = 1/64), and a few ulps beside it
  seams    |y| = |x| and beside it; x tiny against y and y tiny against x (the +-pi/2, 0 and +-pi ends, the u wrap of the texture)
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(n, seed):
    """n uint64 words of the splitmix64 stream started at `seed` (vectorised: word i is the mix of seed + (i+1) * gamma)"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u01(w):       # [0, 1), 53 bits
    return (w >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def sym(w):       # [-1, 1)
    return u01(w) * 2.0 - 1.0


def ulps(x, k):   # x moved by k units in the last place (k integer array, small)
    b = x.view(np.int64) + np.where(x < 0, -k, k).astype(np.int64)
    return b.view(np.float64)


def octants(mn, mx, sel):
    """(y, x) with min(|y|, |x|) = mn, max = mx: sel bit 0 swaps, bits 1 and 2 are the signs"""
    y = np.where(sel & 1, mx, mn)
    x = np.where(sel & 1, mn, mx)
    return np.where(sel & 2, -y, y), np.where(sel & 4, -x, x)


def points():
    out_y, out_x, names = [], [], []

    def add(name, y, x):
        out_y.append(np.asarray(y, np.float64)); out_x.append(np.asarray(x, np.float64)); names.append((name, len(y)))

    n = 1_000_000
    w = splitmix64(3 * n, 1)
    v = np.stack([sym(w[:n]), sym(w[n:2 * n]), sym(w[2 * n:])], 1)
    l = np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])
    keep = (l > 1e-3) & (l <= 1.0)              # uniform directions: rejection from the cube, like point3d.rs:31-38
    v = v[keep] / l[keep, None]
    add("unit", v[:, 0], v[:, 2])

    n = 200_000
    w = splitmix64(2 * n, 2)
    add("plain", sym(w[:n]), sym(w[n:]))

    n = 120_000
    w = splitmix64(4 * n, 3)
    e = lambda q: np.ldexp(1.0, ((q >> np.uint64(20)) % np.uint64(601)).astype(np.int64) - 300)   # noqa: E731
    add("wide", sym(w[:n]) * e(w[n:2 * n]), sym(w[2 * n:3 * n]) * e(w[3 * n:]))

    # knots: mx = m * 2^e with a 40-bit significand m (so mx * k / 32 is exact), mn = mx * k / 32, then a few ulps beside it
    reps = 260
    k = np.tile(np.arange(33, dtype=np.float64), 8 * reps)
    n = k.size
    w = splitmix64(3 * n, 4)
    m = ((w[:n] >> np.uint64(24)) | np.uint64(1 << 39)).astype(np.float64)            # 2^39 .. 2^40, exact
    mx = np.ldexp(m, ((w[n:2 * n] >> np.uint64(30)) % np.uint64(81)).astype(np.int64) - 80)
    mn = mx * k * 0.03125
    d = ((w[2 * n:] >> np.uint64(40)) % np.uint64(7)).astype(np.int64) - 3             # -3 .. 3 ulps
    sel = np.repeat(np.arange(8), 33 * reps)
    y, x = octants(np.where(k > 0, ulps(np.maximum(mn, 5e-324), d), mn), mx, sel)
    add("knots", y, x)

    k = np.tile(np.arange(32, dtype=np.float64), 8 * reps)
    n = k.size
    w = splitmix64(3 * n, 5)
    m = ((w[:n] >> np.uint64(24)) | np.uint64(1 << 39)).astype(np.float64)
    mx = np.ldexp(m, ((w[n:2 * n] >> np.uint64(30)) % np.uint64(81)).astype(np.int64) - 80)
    mn = mx * (2.0 * k + 1.0) * 0.015625                                               # (k + 1/2) / 32, exact
    d = ((w[2 * n:] >> np.uint64(40)) % np.uint64(7)).astype(np.int64) - 3
    sel = np.repeat(np.arange(8), 32 * reps)
    y, x = octants(ulps(mn, d), mx, sel)
    add("halfway", y, x)

    n = 120_000
    w = splitmix64(4 * n, 6)
    a = sym(w[:n]) * np.ldexp(1.0, ((w[n:2 * n] >> np.uint64(20)) % np.uint64(41)).astype(np.int64) - 20)
    a = np.where(a == 0.0, 1.0, a)
    d = ((w[2 * n:3 * n] >> np.uint64(40)) % np.uint64(9)).astype(np.int64) - 4
    kind = (w[3 * n:] >> np.uint64(50)) % np.uint64(4)
    tiny = np.abs(a) * np.ldexp(1.0, -(((w[3 * n:] >> np.uint64(10)) % np.uint64(120)).astype(np.int64) + 20))
    sgn = np.where((w[3 * n:] >> np.uint64(5)) & np.uint64(1), -1.0, 1.0)
    y = np.where(kind == 0, ulps(a, d), np.where(kind == 1, ulps(-a, d), np.where(kind == 2, a, sgn * tiny)))
    x = np.where(kind == 0, a, np.where(kind == 1, a, np.where(kind == 2, sgn * tiny, a)))
    add("seams", y, x)

    return np.concatenate(out_y), np.concatenate(out_x), names
