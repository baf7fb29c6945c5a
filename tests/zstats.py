"""Block z-test between two renders of the same integrand with INDEPENDENT samples (SURVEY §8(d) cross-RNG sanity):
per 8x8 block and channel z = (difference of the block means) / sigma, sigma^2 from the per-pixel variance estimated from
two independent renders (1-dof estimates smoothed over 3x3 blocks).  Under "same distribution" z is standard normal up
to the heavy tails of the variance estimate: mean z ~ 0, mean z^2 ~ 1, |z| > 4 rare; a 1 % brightness bias at these
sample counts gives mean z^2 in the hundreds."""
import numpy as np


def block_z(a_lin, r1_lin, r2_lin, block=8):
    """a: the render under test; r1, r2: two independent reference renders (other seeds).  Returns a dict of statistics."""
    H = (a_lin.shape[0] // block) * block
    W = (a_lin.shape[1] // block) * block
    a, r1, r2 = (x[:H, :W].astype(np.float64) for x in (a_lin, r1_lin, r2_lin))
    var_px = 0.5 * (r1 - r2) ** 2          # unbiased estimate of one render's pixel variance
    diff = a - r1                           # variance 2 var_px under independence

    def blocks(x):
        return x.reshape(H // block, block, W // block, block, 3).mean(axis=(1, 3))
    vb = blocks(var_px) * 2.0 / (block * block)
    pad = np.pad(vb, ((1, 1), (1, 1), (0, 0)), mode="edge")
    vs = sum(pad[i:i + vb.shape[0], j:j + vb.shape[1]] for i in range(3) for j in range(3)) / 9.0
    z = blocks(diff) / np.sqrt(vs + 1e-12)
    noisy = vs > 1e-9                       # (noise-free blocks — plain sky — have no meaningful z)
    zz = z[noisy]
    sig = np.sqrt((2.0 * var_px).sum(axis=(0, 1))) / (H * W)
    return {"n": int(zz.size), "mean_z": float(zz.mean()), "mean_z2": float(np.mean(zz ** 2)), "frac_abs_z_gt_4": float((np.abs(zz) > 4.0).mean()),
            "whole_image_z": (diff.mean(axis=(0, 1)) / np.maximum(sig, 1e-30)).tolist()}


def assert_same_distribution(s, what):
    assert abs(s["mean_z"]) < 0.1 and 0.6 < s["mean_z2"] < 1.6 and s["frac_abs_z_gt_4"] < 5e-3, (what, s)
    assert max(abs(v) for v in s["whole_image_z"]) < 3.5, (what, s)
