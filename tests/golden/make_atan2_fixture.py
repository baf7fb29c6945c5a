#!/usr/bin/env python3
"""Correctly rounded atan2 of tests/atan2_points.py's ~1.1 M argument pairs, by mpmath at 200 bits (the double nearest the
true value; a result within 2^-140 of a rounding boundary would need more, none is).  Writes tests/golden/atan2_cr_low8.npz:
the LOW BYTE of every correctly rounded result (a result one ulp off differs there; that the routine is within a few ulps
is checked against the platform's atan2 separately), the sha256 of the argument bits, and the count per category.

    python tests/golden/make_atan2_fixture.py        (~1 minute)
"""
import hashlib
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from atan2_points import points  # noqa: E402


def main():
    y, x, names = points()
    mp.mp.prec = 200
    out = np.empty(y.size, np.float64)
    for i, (a, b) in enumerate(zip(y.tolist(), x.tolist())):
        out[i] = float(mp.atan2(mp.mpf(a), mp.mpf(b)))     # mpf -> float rounds to nearest even
    low = (out.view(np.uint64) & np.uint64(0xFF)).astype(np.uint8)
    h = hashlib.sha256(y.tobytes() + x.tobytes()).hexdigest()
    np.savez_compressed(os.path.join(HERE, "atan2_cr_low8.npz"), low8=low, args_sha256=np.array(h), categories=np.array([f"{n}:{c}" for n, c in names]))
    print(y.size, "points", names, h)


if __name__ == "__main__":
    main()
