#!/usr/bin/env python3
"""Render one frame on the GPU and print the kernel's counters with derived SIMT-efficiency
ratios (development diagnostics).  python tools/diag.py [--scene S] [--opt key=value ...]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def tail_stats(tl, t0):
    """What the waves do after the tile queue ran dry (RT_PROFILE timeline)."""
    import numpy as np
    qd = (tl[:, 2] - t0).astype(np.float64) / 100.0
    e = (tl[:, 1] - t0).astype(np.float64) / 100.0
    it = (tl[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64)
    ln = (tl[:, 3] >> np.uint64(32)).astype(np.float64)
    ok = it > 0
    q = [0, 10, 50, 90, 100]
    return {"queue_empty_seen_pct": dict(zip(q, np.percentile(qd, q).round(1).tolist())),
            "tail_iters_pct": dict(zip(q, np.percentile(it, q).round(1).tolist())),
            "us_per_tail_iter": round(float(((e - qd)[ok]).sum() / it[ok].sum()), 2),
            "us_per_tail_iter_longest10": round(float(np.sort((e - qd))[-10:].sum() / it[np.argsort(e - qd)[-10:]].sum()), 2),
            "mean_lanes_in_tail_iter": round(float(ln.sum() / max(1.0, it.sum())), 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="scenes/cfg2_cover_1200x800_spp128.json")
    ap.add_argument("--opt", nargs="*", default=[])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--lib", default="", help="alternative librt_hip .so (an A/B build)")
    ap.add_argument("--shard", default="", help="rank,world: render only that rank's interleaved 8-row tiles (multi-GPU emulation)")
    ap.add_argument("--procedural", type=int, default=0, help="use the procedural world with this `half` (50 -> ~10 000 spheres)")
    ap.add_argument("--radii", default="uniform", help="procedural world: uniform | loguniform | bimodal (scenes/procedural.py)")
    a = ap.parse_args()
    import torch
    os.chdir(ROOT)
    pkg = graft.load_package()
    if a.lib:   # (a -DRT_PROFILE -DRT_TEST_PROBES build: the timeline comes through rt_hip_debug_timeline, include/rt_abi_test.h)
        pkg.hip.LIB_PATH = pkg.hip.PROBE_LIB_PATH = os.path.abspath(a.lib)
    if a.procedural:
        sys.path.insert(0, os.path.join(ROOT, "scenes"))
        import procedural
        sc = pkg.host.Scene.loads(procedural.make_json(width=a.width or 3840, height=a.height or 2160, spp=a.spp or 8, half=a.procedural, seed=0, radii=a.radii))
        a.scene = f"procedural_half{a.procedural}" + ("" if a.radii == "uniform" else "_" + a.radii)
    else:
        sc = pkg.host.Scene.load(a.scene)
    if a.width:
        sc.c.width = a.width
    if a.height:
        sc.c.height = a.height
    if a.spp:
        sc.c.samples_per_pixel = a.spp
    gs = pkg.hip.HipScene(sc.ptr, 0, library=pkg.hip.probe_lib() if a.lib else None)   # (--lib: a profile build with the debug calls)
    for kv in a.opt:
        k, v = kv.split("=")
        gs.set_option(k, int(v))
    rgb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
    tiles = None
    if a.shard:
        parts = [int(v) for v in a.shard.split(",")]
        r, w = parts[0], parts[1]
        tiles = pkg.abi.RtRowTiles(parts[2] if len(parts) > 2 else 8, r, w)
    best, timeline = None, None
    for _ in range(a.reps):
        gs.render(rgb.data_ptr(), 0, tiles, torch.cuda.current_stream().cuda_stream)
        st = gs.wait()
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
            if st["prof_cycles"][6]:
                timeline, raw_counters = gs.debug_timeline()
    st = best
    wi = st["wave_iters"]
    out = dict(scene=os.path.basename(a.scene), opts=a.opt, kernel_ms=round(st["kernel_ms"], 3),
               msamples_per_s=round(st["samples"] / st["kernel_ms"] / 1e3, 1),
               segments_per_sample=round(st["segments"] / max(1, st["samples"]), 3),
               exact_per_segment=round(st["exact_tests"] / max(1, st["segments"]), 3),
               steps_per_segment=round(st["grid_steps"] / max(1, st["segments"]), 3),
               wave_iters=wi,
               lane_util_segments=round(st["segments"] / max(1, 64 * wi[0]), 4),
               wave_step_iters_per_wave_iter=round(wi[1] / max(1, wi[0]), 3),
               wave_test_iters_per_wave_iter=round(wi[2] / max(1, wi[0]), 3),
               lane_util_steps=round(st["grid_steps"] / max(1, 64 * wi[1]), 4),
               lane_util_tests=round(st["exact_tests"] / max(1, 64 * wi[2]), 4),
               items=wi[3])
    pc = st["prof_cycles"]
    if pc[6]:
        names = ["refill", "large", "lane_shade", "walk", "accumulate", "item", "total"]
        out["prof_share"] = {n: round(pc[i] / pc[6], 4) for i, n in enumerate(names[:6])}
        if pc[7]:
            out["wave_us_at_2.4GHz"] = {"longest": round(pc[7] / 2400.0, 1), "shortest": round(pc[8] / 2400.0, 1),
                                        "mean": round(pc[6] / max(1, pc[10]) / 2400.0, 1), "waves": pc[10]}
        if timeline is not None and len(timeline):
            import numpy as np
            t0 = timeline[:, 0].min()
            b = (timeline[:, 0] - t0).astype(np.float64) / 100.0   # us
            e = (timeline[:, 1] - t0).astype(np.float64) / 100.0
            q = [0, 1, 10, 50, 90, 99, 100]
            out["timeline_us"] = {"start_pct": dict(zip(q, np.percentile(b, q).round(1).tolist())),
                                  "end_pct": dict(zip(q, np.percentile(e, q).round(1).tolist())),
                                  "idle_frac_after_end": round(float((e.max() - e).sum() / (len(e) * e.max())), 4),
                                  "tail": tail_stats(timeline, t0),
                                  "tail_section_share": dict(zip(["refill", "large", "lane_shade", "walk", "accumulate"],
                                                                 (raw_counters[19:24] / max(1.0, float(raw_counters[19:24].sum()))).round(3).tolist())),
                                  "tail_cycles_per_iter": round(float(raw_counters[19:24].sum()) / max(1.0, float((timeline[:, 3] & np.uint64(0xFFFFFFFF)).sum())), 1),
                                  "end_by_xcd_mean": [round(float(e.reshape(-1, 16)[x::8].mean()), 1) for x in range(8)]}
        out["wave_busy_frac_at_2.4GHz"] = round(pc[6] / (max(1, pc[10]) * st["kernel_ms"] * 2.4e6), 3)
        out["prof_cycles_per_wave_iter"] = {n: round(pc[i] / max(1, wi[0]), 1) for i, n in enumerate(names)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
