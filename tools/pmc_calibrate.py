#!/usr/bin/env python3
"""Calibrate rocprofv3's WRITE_SIZE / FETCH_SIZE on THIS box against known byte counts (MI355X_MICROARCH.md §HBM: "WRITE_SIZE
is uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").

    cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/pmc_w -o d -- python tools/pmc_calibrate.py
    cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/pmc_f -o d -- python tools/pmc_calibrate.py
    python tools/pmc_calibrate.py --report <dir>

Patterns (each a single torch kernel launch, sizes far beyond L2):
  fill      256 MiB of dwordx4 stores            -> writes 256 MiB
  copy      256 MiB read + 256 MiB written
  bytes/16  one byte stored in every 16-byte chunk of 256 MiB (strided u8 fill): how PARTIAL writes are counted — the
            framebuffer leaves the megakernel as 6- and 12-byte pieces"""
import csv
import glob
import json
import sys


def run():
    import torch
    n = 256 << 20
    a = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    b = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(3):
        a.fill_(7)            # fill
        torch.cuda.synchronize()
        b.copy_(a)            # copy
        torch.cuda.synchronize()
        a[::16].fill_(9)      # one byte per 16
        torch.cuda.synchronize()
        a.view(torch.int32)[::4].fill_(5)   # one dword per 16 bytes
        torch.cuda.synchronize()
    print("calibration kernels done")


def report(d):
    rows = []
    for f in glob.glob(d + "/pmc_*/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    out = {}
    for r in rows:
        k = r["Kernel_Name"]
        if "Fill" not in k and "copy" not in k.lower() and "elementwise" not in k:
            continue
        key = (r["Counter_Name"], k[:90], r.get("Grid_Size", ""))
        out.setdefault(key, []).append(float(r["Counter_Value"]))
    for (c, k, g), v in sorted(out.items()):
        print(f"{c:11s} mean {sum(v) / len(v) / 1024.0:10.1f} MiB over {len(v)} launches  grid {g:>10s}  {k}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
