"""A SECOND, independent restatement of the reference's ray_color path — a check on the first (oracle/rt_oracle.c).

TEST INFRASTRUCTURE.  Written from /root/reference/raytracer/src/*.rs alone (raytracer.rs:44-165, 191-218; sphere.rs:35-79;
materials.rs:65-69, 84-95, 111-129, 144-155, 176-199, 236-267; point3d.rs:22-86; camera.rs:79-84; ray.rs:18-20) WITHOUT opening
rt_oracle.c: plain Python floats for the f64 geometry (CPython never fuses a*b+c), numpy.float32 scalars for colour, recursion
as in the reference.  What it shares with the first restatement is only what the reference does not define: the RNG — Philox4x32-10
by (pixel, sample, node, slot) as rt_core.h:292-323 / DESIGN.md §2 specify it (restated here in Python, pinned by the Random123
vectors in tests/test_oracle_kat.py) — the nest cap of 8, the texel clamp, and the one correctly rounded atan2 (rt_atan2.h,
through the oracle library's rt_oracle_atan2 hook).  tests/test_oracle_kat.py::test_second_restatement_* asserts that the two
restatements produce the SAME BITS (linear radiance, RGB8) and the same segment counts on lit / textured / glass frames.
"""
import math

import numpy as np

F = np.float32
M32 = 0xFFFFFFFF
NODE_CAMERA = 0xFFFFFFFF
MAX_LIGHT_NEST = 8
LAMBERTIAN, METAL, GLASS, TEXTURE, LIGHT = range(5)
EPS = 2.220446049250313e-16   # f64::EPSILON
F64_MAX = 1.7976931348623157e308


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def u01_53(lo, hi):            # rand 0.8 Standard f64: (u64 >> 11) * 2^-53
    return float(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)


def range_m1_1(w):             # gen_range(-1.0..1.0) on the 2^-32 grid: v * (hi - lo) + lo
    return (w * (1.0 / 4294967296.0)) * 2.0 + (-1.0)


def child_node(node, j):       # RNG node of light ray j shot from a hit whose node is `node` (rt_core.h:319-323)
    x = (node * 0x9E3779B1 + (j + 1) * 0x85EBCA77) & M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    return 0x80000000 | (x & 0x7FFFFFFE)


def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def muls(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def divs(a, s): return (a[0] / s, a[1] / s, a[2] / s)
def neg(a): return (-a[0], -a[1], -a[2])
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def len2(a): return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]


def sqrt(x):                   # f64::sqrt: NaN for a negative argument (math.sqrt raises)
    return math.sqrt(x) if x >= 0.0 else float("nan")


def length(a):                 # point3d.rs:52-65: distance to the origin
    dx, dy, dz = a[0] - 0.0, a[1] - 0.0, a[2] - 0.0
    return math.sqrt(dx * dx + dy * dy + dz * dz)


def unit(a):
    l = length(a)
    return (a[0] / l, a[1] / l, a[2] / l)


def clamp(v):                  # raytracer.rs:61-69 (NaN passes)
    return F(0.0) if v < F(0.0) else (F(1.0) if v > F(1.0) else v)


def trunc_usize(x):            # Rust `as usize`: truncates, saturates, NaN -> 0
    x = float(x)
    return 0 if (x != x or x <= 0.0) else (2 ** 64 - 1 if x >= 18446744073709551616.0 else int(x))


class Mini:
    def __init__(self, scene, atan2):
        sc = self.sc = scene
        self.atan2 = atan2
        self.obj = [sc.spheres[i] for i in range(sc.n_spheres)]
        self.geom = [(tuple(o.center), o.radius) for o in self.obj]
        self.lights = [i for i, o in enumerate(self.obj) if o.kind == LIGHT]          # raytracer.rs:220-229
        self.tex = [np.ctypeslib.as_array(sc.textures[t].rgb8, (sc.textures[t].nbytes,)) for t in range(sc.n_textures)]
        self.sky = np.ctypeslib.as_array(sc.sky_rgb8, (sc.sky_w * sc.sky_h * 3,)) if sc.sky_mode == 2 else None
        self.k0, self.k1 = sc.seed & M32, (sc.seed >> 32) & M32
        self.segments = 0

    def words(self, node, slot):
        return philox4x32_10(self.pixel, self.sample, node, slot, self.k0, self.k1)

    def random_in_unit_sphere(self, node):       # point3d.rs:22-38
        a = 0
        while True:
            w = self.words(node, 1 + a)
            p = (range_m1_1(w[0]), range_m1_1(w[1]), range_m1_1(w[2]))
            if len2(p) < 1.0:
                return p
            a += 1

    def hit_world(self, o, d):                   # raytracer.rs:44-59 + sphere.rs:46-78
        closest, best = F64_MAX, None
        a = len2(d)
        for i, (c, r) in enumerate(self.geom):
            oc = sub(o, c)
            half_b = dot(oc, d)
            cc = len2(oc) - r * r
            disc = (half_b * half_b) - (a * cc)
            if disc >= 0.0:
                sq = math.sqrt(disc)
                for root in (((-half_b) - sq) / a, ((-half_b) + sq) / a):
                    if root < closest and root > 0.001:
                        closest, best = root, i
                        break
        if best is None:
            return None
        c, r = self.geom[best]
        p = add(o, muls(d, closest))
        normal = divs(sub(p, c), r)
        front = dot(d, normal) < 0.0
        return best, p, (normal if front else neg(normal)), front

    def texel(self, o, p):                       # sphere.rs:35-43 + materials.rs:236-254
        c = tuple(o.center)
        n = unit(sub(p, c))
        u = (self.atan2(n[0], n[2]) / (2.0 * math.pi)) + 0.5
        v = n[1] * 0.5 + 0.5
        rot = u + o.h_offset
        if rot > 1.0:
            rot = rot - 1.0
        uu, vv = rot * float(o.tex_w), (1.0 - v) * float(o.tex_h - 1)
        px = self.tex[o.tex_id]
        base = 3 * (trunc_usize(math.floor(vv)) * o.tex_w + trunc_usize(math.floor(uu)))
        base = min(base, len(px) - 3)            # (the reference panics beyond the buffer: clamp, shared deviation 4)
        return tuple(F(px[base + k]) / F(255.0) for k in range(3))

    def scatter(self, i, d, p, n, front, node):  # -> None | (direction | None, albedo f32x3)
        o = self.obj[i]
        if o.kind == LIGHT:
            return None, (F(1.0), F(1.0), F(1.0))
        if o.kind in (LAMBERTIAN, TEXTURE):
            sd = add(n, self.random_in_unit_sphere(node))
            if abs(sd[0]) < EPS and abs(sd[1]) < EPS and abs(sd[2]) < EPS:
                sd = n
            alb = tuple(F(x) for x in o.albedo) if o.kind == LAMBERTIAN else self.texel(o, p)
            return sub(add(p, sd), p), alb
        if o.kind == METAL:
            refl = sub(d, muls(n, 2.0 * dot(d, n)))
            sd = add(refl, muls(self.random_in_unit_sphere(node), o.fuzz_or_ior))
            return (sd, tuple(F(x) for x in o.albedo)) if dot(sd, n) > 0.0 else None
        ratio = 1.0 / o.fuzz_or_ior if front else o.fuzz_or_ior   # Glass, materials.rs:176-199
        ud = unit(d)
        cos_t = min(dot(neg(ud), n), 1.0)
        sin_t = sqrt(1.0 - cos_t * cos_t)
        r0 = (1.0 - ratio) / (1.0 + ratio)
        r0 = r0 * r0
        x = 1.0 - cos_t
        x2 = x * x
        reflectance = r0 + (1.0 - r0) * (x2 * x2 * x)            # powi(5): x^2, x^4, x^4 * x
        w = self.words(node, 0)
        one = (F(1.0), F(1.0), F(1.0))
        if ratio * sin_t > 1.0 or reflectance > u01_53(w[0], w[1]):
            return sub(ud, muls(n, 2.0 * dot(ud, n))), one
        cos2 = min(dot(neg(ud), n), 1.0)                          # refract(), materials.rs:144-149
        perp = muls(add(ud, muls(n, cos2)), ratio)
        par = muls(n, -1.0 * math.sqrt(abs(1.0 - len2(perp))))
        return add(perp, par), one

    def sky_colour(self, d):                     # raytracer.rs:134-162
        sc = self.sc
        ud = unit(d)
        t, u = clamp(F(0.5) * (F(ud[1]) + F(1.0))), clamp(F(0.5) * (F(unit(d)[0]) + F(1.0)))
        if sc.sky_mode == 0:
            return F(0.0), F(0.0), F(0.0)
        if sc.sky_mode == 1:
            return tuple((F(1.0) - t) * F(1.0) + t * F(k) for k in (0.5, 0.7, 1.0))
        x, y = trunc_usize(u * F(sc.sky_w - 1)), trunc_usize((F(1.0) - t) * F(sc.sky_h - 1))
        return tuple(F(0.7) * F(self.sky[(y * sc.sky_w + x) * 3 + k]) / F(255.0) for k in range(3))

    def ray_color(self, o, d, max_depth, depth, node, nest):     # raytracer.rs:71-165
        if depth <= 0:
            return F(0.0), F(0.0), F(0.0)
        self.segments += 1
        hit = self.hit_world(o, d)
        if hit is None:
            return self.sky_colour(d)
        i, p, n, front = hit
        sc = self.scatter(i, d, p, n, front, node)
        if sc is None:
            return F(0.0), F(0.0), F(0.0)
        sdir, alb = sc
        light = [F(0.0), F(0.0), F(0.0)]
        nl = len(self.lights)
        prob = 0.05 if self.obj[i].kind == GLASS else 0.1
        if nl > 0:
            w0 = self.words(node, 0)
            draw = u01_53(w0[2], w0[3]) if self.obj[i].kind == GLASS else u01_53(w0[2], self.words(node, 1)[3])
            if draw > (1.0 - float(nl) * prob) and depth > ((max_depth - 2) % 2 ** 64) and nest < MAX_LIGHT_NEST:
                for j, li in enumerate(self.lights):
                    tc = self.ray_color(p, sub(tuple(self.obj[li].center), p), 2, 1, child_node(node, j), nest + 1)
                    for k in range(3):
                        light[k] = light[k] + alb[k] * tc[k]
                light = [x / F(nl) for x in light]
        if sdir is None:
            return alb
        tc = self.ray_color(p, sdir, max_depth, depth - 1, node + 1, nest)
        return tuple(clamp(light[k] + alb[k] * tc[k]) for k in range(3))

    def render(self):                            # raytracer.rs:191-218
        sc = self.sc
        W, H, spp = sc.width, sc.height, sc.samples_per_pixel
        org, ll, hor, ver = (tuple(v) for v in (sc.cam_origin, sc.cam_lower_left, sc.cam_horizontal, sc.cam_vertical))
        lin, rgb = np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.uint8)
        for y in range(H):
            for x in range(W):
                acc = [F(0.0), F(0.0), F(0.0)]
                self.pixel = y * W + x
                for s in range(spp):
                    self.sample = s
                    w = self.words(NODE_CAMERA, 0)
                    u = (float(x) + u01_53(w[0], w[1])) / (float(W) - 1.0)
                    v = (float(H) - (float(y) + u01_53(w[2], w[3]))) / (float(H) - 1.0)
                    d = sub(add(add(ll, muls(hor, u)), muls(ver, v)), org)       # camera.rs:79-84
                    c = self.ray_color(org, d, sc.max_depth, sc.max_depth, 0, 0)
                    acc = [acc[k] + c[k] for k in range(3)]
                scale = F(1.0) / F(spp)
                for k in range(3):
                    lin[y, x, k] = scale * acc[k]
                    g = np.sqrt(scale * acc[k]) * F(255.0)       # palette into_format: round(min(255 x, 255)), negatives -> 0
                    rgb[y, x, k] = 255 if g != g else int(np.rint(min(max(g, F(0.0)), F(255.0))))
        return rgb, lin, self.segments
