#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/gpu_check.sh, tools/pmc_scene.sh): mean counter value per launch of
the megakernel, plus HBM bytes per launch as MI355X_MICROARCH.md (HBM section) prescribes:
(FETCH_SIZE + WRITE_SIZE) * 1024, with FETCH_SIZE doubled (gfx950 reports half the bytes of a
wide read stream; an upper bound for other access widths).  Output is profiles-ready, SELF-DESCRIBING JSON:
the scene, the kernel time the passes themselves measured (`kernel_ms`, from the JSON lines bench.py / diag.py
printed into the passes' logs) and the commit the library was built from (`git_head`, BUILD_INFO.json).

    python tools/pmc_summary.py <dir with pmc_*/ subdirectories and *.log> [scene label]"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
if len(sys.argv) > 2:
    summary["scene"] = sys.argv[2]
# kernel time of the runs the counters come from (the JSON lines in the passes' logs)
kms = []
for f in glob.glob(out + "/*.log"):
    for line in open(f, errors="replace"):
        line = line.strip()
        if line.startswith("{") and '"kernel_ms"' in line:
            try:
                kms.append(float(json.loads(line)["kernel_ms"]))
            except Exception:
                pass
if kms:
    summary["kernel_ms"] = round(sum(kms) / len(kms), 4)
    summary["kernel_ms_note"] = f"mean over the {len(kms)} counter passes (HIP events; a pass with counters on can run slower than an unprofiled one)"
try:
    info = json.load(open(os.path.join(ROOT, "rust-raytracer_amd", "BUILD_INFO.json")))
    summary["git_head"] = info["git_head"]
    summary["kernel_src_hash"] = info.get("kernel_src_hash")
except Exception:
    summary["git_head"] = None
# TCC FETCH_SIZE / WRITE_SIZE count kilobytes (rocprof derived counters).  On gfx950 FETCH_SIZE can
# under-report reads by up to 2x (MI355X_MICROARCH.md, HBM section): report raw and corrected.
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    fetch_b, write_b = res["FETCH_SIZE"] * 1024.0, res["WRITE_SIZE"] * 1024.0
    summary["hbm_bytes_per_launch_raw"] = fetch_b + write_b
    summary["hbm_bytes_per_launch"] = 2.0 * fetch_b + write_b
    summary["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, "
                         "FETCH_SIZE x2 (gfx950 under-report upper bound)")
if "SQ_THREAD_CYCLES_VALU" in res and "SQ_ACTIVE_INST_VALU" in res:
    summary["lane_utilisation"] = round(res["SQ_THREAD_CYCLES_VALU"] / (64.0 * res["SQ_ACTIVE_INST_VALU"]), 4)
    if kms:
        k = summary["kernel_ms"] * 1e-3
        summary["lane_slot_frac_at_that_time"] = round(res["SQ_THREAD_CYCLES_VALU"] / (256 * 4 * 16 * 2.4e9 * k), 4)
if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res and res["TCC_HIT_sum"] + res["TCC_MISS_sum"] > 0:
    summary["l2_hit_rate"] = round(res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"]), 4)
print(json.dumps(summary, indent=1))
