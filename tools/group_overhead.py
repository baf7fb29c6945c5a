#!/usr/bin/env python3
"""Where an in-library multi-GPU frame spends its NON-kernel time, measured on ONE GPU:

  * G = 1 with RT_GATHER_SELFTEST=1 (the whole path: launch, ONE gather — RCCL communicator of one rank, or the peer leg —,
    de-interleave, event waits) on a 1200 x 100 frame = the work of one rank of the 8-GPU headline frame;
  * G = 8 with RT_GPUS_EMULATE=1 (eight host threads, eight launches, seven peer copies on one device: the kernels
    serialise, but the HOST-side stage clocks — thread wake-up, enqueue, submit — are what eight real devices would see);
  * blocking frames (frame_ms - kernel_ms, RtStats.group_us stage table) and pipelined frames (ms per frame against the
    slowest rank's kernel), with the rank threads polling (spin_us) or sleeping between frames.

    python tools/group_overhead.py [--reps N]  > gpurun_out/group_overhead.json   (one JSON object per line)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

STAGES = ("last_rank_thread_running", "last_rank_enqueued", "submitter_knows", "gather_enqueued", "submit_returns", "assembled_seen", "frame_done", "stats_read")


def med(v):
    return sorted(v)[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    import numpy as np
    os.chdir(ROOT)
    pkg = graft.load_package()
    cases = [("G=1 selftest rccl, 1200x100 (one rank's work of the 8-GPU frame)", 1, {"RT_GATHER_SELFTEST": "1", "RT_GATHER": "rccl"}, 100),
             ("G=1 selftest peer, 1200x100", 1, {"RT_GATHER_SELFTEST": "1", "RT_GATHER": "peer"}, 100),
             ("G=1 no gather, 1200x100", 1, {}, 100),
             ("G=8 emulated on one device (peer), 1200x800", 8, {"RT_GPUS_EMULATE": "1"}, 800),
             ("G=1 selftest rccl, 1200x800 whole frame", 1, {"RT_GATHER_SELFTEST": "1", "RT_GATHER": "rccl"}, 800)]
    for name, G, env, height in cases:
        sc = pkg.host.Scene.load("scenes/cfg2_cover_1200x800_spp128.json")
        sc.c.height = height   # (the camera's aspect stays the headline frame's: these are its top rows, sky — the cheapest work, the overhead shows most)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        sys.stdout.flush()
        saved = os.dup(1); os.dup2(2, 1)      # RCCL's banner goes to stderr
        try:
            grp = pkg.hip.HipGroup(sc.ptr, G)
        finally:
            os.dup2(saved, 1); os.close(saved)
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        host = np.zeros((height, 1200, 3), np.uint8)
        for spin in (0, 3000):
            grp.set_option("spin_us", spin)
            for _ in range(4):
                grp.render()
            rows = {"blocking_in_hbm": [], "blocking_to_host": []}
            for _ in range(a.reps):
                rows["blocking_in_hbm"].append(grp.render())
                rows["blocking_to_host"].append(grp.render_to_host(host)[1])
            rec = {"case": name, "n_ranks": G, "spin_us": spin, "transport": grp.info()["transport"]}
            for key, sts in rows.items():
                rec[key] = {"frame_ms": round(med([s["frame_ms"] for s in sts]), 4), "kernel_ms": round(med([s["kernel_ms"] for s in sts]), 4),
                            "non_kernel_ms": round(med([s["frame_ms"] - s["kernel_ms"] for s in sts]), 4),
                            "gather_ms_device_clock": round(med([s["gather_ms"] for s in sts]), 4),
                            "stages_us": {k: round(med([s["group_us"][i] for s in sts]), 1) for i, k in enumerate(STAGES)}}
            # pipelined: K frames, two in flight
            K = a.reps
            t0 = time.perf_counter()
            grp.submit()
            ks = []
            for _ in range(K - 1):
                grp.submit()
                ks.append(grp.collect()["kernel_ms"])
            ks.append(grp.collect()["kernel_ms"])
            el = (time.perf_counter() - t0) * 1e3 / K
            rec["pipelined"] = {"ms_per_frame": round(el, 4), "kernel_ms": round(med(ks), 4), "non_kernel_ms": round(el - med(ks), 4)}
            t0 = time.perf_counter()
            grp.submit(host)
            for _ in range(K - 1):
                grp.submit(host)
                grp.collect()
            grp.collect()
            rec["pipelined_to_host"] = {"ms_per_frame": round((time.perf_counter() - t0) * 1e3 / K, 4)}
            print(json.dumps(rec), flush=True)
        grp.close()


if __name__ == "__main__":
    main()
