"""The parity bar between the HIP megakernel (or its CPU build, tests/hostsim) and the oracle.

Geometry (hit decisions, hit points, scatter directions, RNG) is bit-identical f64 on both
sides, so both trace exactly the same paths.  Colour is f32: the oracle multiplies the
per-level attenuations innermost-first (the recursion of raytracer.rs:117-122), the kernel
carries them outermost-first (rt_core.h `Fwd`), so a sample can differ by a few f32 ulps.

  LINEAR_ATOL  per-channel |difference| of the mean linear radiance (before sqrt gamma)
  RGB8         identical, except at most RGB8_FLIP_FRAC of the values may differ by 1 LSB
               (a 1e-7 difference straddling a rounding boundary of x*255)
"""
import numpy as np

LINEAR_ATOL = 2e-6
RGB8_FLIP_FRAC = 1e-4


def pooled_atol(spp):
    """The default (pooled-sample) kernel sums a pixel's samples in exact fixed point instead of
    the reference's sequential f32 adds (raytracer.rs:203-205).  The two differ by the f32
    sum's own rounding, bounded by ~spp * 2^-25 on the mean: measured 1.3e-6 at spp 128."""
    return LINEAR_ATOL + 3e-8 * spp


def assert_parity(got_rgb, got_lin, want_rgb, want_lin, what="", atol=LINEAR_ATOL, flip_frac=RGB8_FLIP_FRAC):
    assert got_rgb.shape == want_rgb.shape and got_lin.shape == want_lin.shape, what
    assert np.isfinite(got_lin).all(), what
    err = float(np.abs(got_lin.astype(np.float64) - want_lin.astype(np.float64)).max())
    assert err <= atol, f"{what}: max |linear diff| {err:.3g} > {atol}"
    d = np.abs(got_rgb.astype(np.int16) - want_rgb.astype(np.int16))
    assert d.max() <= 1, f"{what}: RGB8 differs by {d.max()} LSB"
    flips = int((d != 0).sum())
    assert flips <= max(2, int(flip_frac * d.size)), f"{what}: {flips} of {d.size} RGB8 values differ"
    return err, flips
