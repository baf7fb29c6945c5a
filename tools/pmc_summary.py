#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/gpu_check.sh: mean counter value per launch of
the megakernel, plus HBM bytes per launch as MI355X_MICROARCH.md (HBM section) prescribes:
(FETCH_SIZE + WRITE_SIZE) * 1024, with FETCH_SIZE doubled (gfx950 reports half the bytes of a
wide read stream; an upper bound for other access widths).  Output is profiles-ready JSON."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_megakernel" not in r.get("Kernel_Name", ""):
            continue
        a = agg[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1].add((f, r.get("Dispatch_Id")))
res = {k: v[0] / max(1, len(v[1])) for k, v in agg.items()}
summary = {"kernel": "rt_megakernel", "mean_per_launch": res}
# TCC FETCH_SIZE / WRITE_SIZE count kilobytes (rocprof derived counters).  On gfx950 FETCH_SIZE can
# under-report reads by up to 2x (MI355X_MICROARCH.md, HBM section): report raw and corrected.
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    fetch_b, write_b = res["FETCH_SIZE"] * 1024.0, res["WRITE_SIZE"] * 1024.0
    summary["hbm_bytes_per_launch_raw"] = fetch_b + write_b
    summary["hbm_bytes_per_launch"] = 2.0 * fetch_b + write_b
    summary["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, "
                         "FETCH_SIZE x2 (gfx950 under-report upper bound)")
print(json.dumps(summary, indent=1))
