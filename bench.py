#!/usr/bin/env python3
"""bench.py — Msamples/s of the ray_color hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one frame of BASELINE configs[1] — cover_scene at 1200x800, spp 128, depth 50,
484 spheres — rendered by the HIP megakernel through the C ABI (librt_hip.so), with the
scene tables already resident in HBM.  With N > 1 the frame is sharded by interleaved
2-scanline tiles (rank r renders tiles r, r+N, ...) and assembled on rank 0 by ONE gather
over RCCL; total work is fixed, so scaling is "strong" (the north-star target is a >=6x
speed-up of this frame at 8 GPUs).

Two hosts for N > 1, the line's config.parallelism says which ran:
  * `python bench.py --gpus N` (no torchrun): ONE process drives the PRODUCT path — the in-library group
    (rt_hip_group_*: host thread + stream per device, ncclCommInitAll, one ncclGather per frame, de-interleave
    kernel), K frames pipelined two deep (rt_hip_group_submit / _collect: frame i's gather under frame i+1's kernels), each
    assembled in HBM of the first device; `value` from wall time.  Then blocking frames one at a time: `frame_latency_ms`,
    its non-kernel part and the host-clock stage table (`frame_latency_stages_us`).  RT_GPUS_EMULATE=1 lets the ranks share
    devices (a 1-GPU box can run the whole path).
  * under torch.distributed.run (WORLD_SIZE set): one process per GPU, the same shards gathered by
    torch.distributed (RCCL), described next.
Any failure prints ONE JSON line with an "error" key and exits non-zero — never a bare traceback.

Process-per-GPU branch: in the timed region the tile buffers are double-buffered:
frame i's gather runs (asynchronously, on the collective's stream) while frame i+1 is being
rendered, and all K frames are assembled on rank 0 before the closing barrier — `value` is that
pipelined throughput.  The latency of ONE frame with nothing overlapped (render, gather, row
permutation on rank 0) is measured separately after the timed region: `frame_latency_ms`, next
to `n1_kernel_ms` (rank 0 rendering the whole frame alone) and their ratio.
Rank 0 prints ONE JSON line.

Extra objects on that line (N = 1):
  roofline      the megakernel against the roof that binds it: vector-ALU issue (no dense contraction -> no MFMA;
                ~10 MB of HBM traffic per frame -> not HBM).  `achieved` = executed VALU lane-slots per second =
                SQ_THREAD_CYCLES_VALU per launch (rocprofv3 --pmc, profiles/<round>_pmc.json, same kernel build) /
                this run's average kernel time (HIP events on the launch stream); `peak` = 256 CUs x 4 SIMDs x 16
                lanes x 2.4 GHz = 39.32 T lane-slots/s (x 2 flop per FMA = the 78.6 TFLOP/s FP64 vector peak);
                `frac` = their ratio = (VALU issue busy) x (lane utilisation), both restated from the counters.
                `executed_f64`: exact Sphere::hit tests actually run x 17 flop / kernel time vs 78.6 TFLOP/s.
                `algorithmic`: the reference's brute force (segments x n_spheres tests) over the same
                time — a speed-up figure (the grid walk runs ~1 % of those tests), NOT a roofline fraction.
                `traffic`: HBM bytes per launch from the PMC passes, and its ratio to the algorithmic bytes.
  cpu_baseline  the CPU oracle (a literal restatement of the reference's rayon path) timed
                on this box's host cores on a bounded sample of the same frame
  other_configs the other BASELINE configs at full size on one GPU (2 frames each; median of 20 for frames under 50 ms): kernel_ms, Msamples/s
"""
import argparse
import glob
import re
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HEADLINE = "scenes/cfg2_cover_1200x800_spp128.json"
FLOP_PER_TEST = 17          # sphere.rs:47-53 with |d|^2 and r^2 hoisted
PEAK_FP64_VALU_TFLOPS = 78.6    # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PEAK_LANE_SLOTS_T = 256 * 4 * 16 * 2.4e9 / 1e12   # 39.32 T VALU lane-slots/s
ENGINE_HZ = 2.4e9
N_SIMD = 256 * 4


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _latest_pmc():
    """newest profiles/rNN_*pmc.json that carries the VALU counters of the headline kernel"""
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
        try:
            j = json.load(open(p))
        except Exception:
            continue
        m = j.get("mean_per_launch", {})
        if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
            b = os.path.basename(p)
            run = re.search(r"run(\d+)", b)
            key = (b.split("_")[0], int(run.group(1)) if run else -1, os.path.getmtime(p))
            if best is None or key > best[0]:
                best = (key, p, j)
    return (best[1], best[2]) if best else (None, None)


def _args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default=HEADLINE, help="scene JSON (default: BASELINE configs[1])")
    ap.add_argument("--spp", type=int, default=0, help="override samples_per_pixel (non-headline run)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-row-stride", type=int, default=16,
                    help="cpu_baseline renders every k-th scanline (default: chosen for ~10 s of CPU wall time)")
    return ap.parse_args()


class BenchError(RuntimeError):
    pass


def _pmc_head(pmc):
    return (pmc or {}).get("git_head")


def _build_info(key):
    try:
        return json.load(open(os.path.join(ROOT, "rust-raytracer_amd", "BUILD_INFO.json"))).get(key)
    except Exception:
        return None


def _git_head():
    """the commit librt_hip.so was built from (BUILD_INFO.json, written by build(): the GPU box has no .git)"""
    try:
        return json.load(open(os.path.join(ROOT, "rust-raytracer_amd", "BUILD_INFO.json"))).get("git_head")
    except Exception:
        return None


def _scene_pmc(key):
    """newest profiles/rNN_*pmc_<key>.json (counters of another BASELINE config, tools/pmc_scene.sh)"""
    def run_key(p):   # (round, run number): "r03_run11" is newer than "r03_run1" — a plain string sort says otherwise
        b = os.path.basename(p)
        run = re.search(r"run(\d+)", b)
        return (b.split("_")[0], int(run.group(1)) if run else -1)
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*pmc_{key}.json")), key=run_key)
    if not c:
        return None
    try:
        return c[-1], json.load(open(c[-1]))
    except Exception:
        return None


def main():
    args = _args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    try:
        if world == 1 and args.gpus > 1:
            line = run_group(args)          # ONE process, the in-library group (the product's multi-GPU path)
        else:
            line = run_ranks(args)          # N = 1, or one process per GPU under torch.distributed.run
    except BaseException as e:              # noqa: BLE001 — the driver must get a JSON line whatever happens
        if isinstance(e, KeyboardInterrupt):
            raise
        if rank == 0:
            import traceback
            traceback.print_exc(file=sys.stderr)
            _flush_c_stdio()
            print(json.dumps({"metric": "Msamples/sec (pixels x spp / s) on cover_scene 1200x800", "value": None, "unit": "Msamples/s",
                              "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": f"{type(e).__name__}: {e}"}), flush=True)
        sys.exit(2)
    _flush_c_stdio()
    if rank == 0 and line is not None:
        print(line, flush=True)   # the ONE JSON line, and the last thing this process writes to stdout


def run_group(args):
    """--gpus N without torchrun: the in-library group (rt_hip_group_*), one process: pipelined frames, then blocking ones."""
    import numpy as np
    import torch

    os.chdir(ROOT)
    if not torch.cuda.is_available():
        raise BenchError("bench.py needs a GPU: the hot path has no CPU fallback")
    pkg = graft.load_package()
    abi, host, hip = pkg.abi, pkg.host, pkg.hip
    visible = hip.device_count()
    emulate = os.environ.get("RT_GPUS_EMULATE") == "1"
    if args.gpus > visible and not emulate:
        raise BenchError(f"--gpus {args.gpus} but only {visible} device(s) visible (RT_GPUS_EMULATE=1 lets ranks share devices: tests only)")
    sc = host.Scene.load(args.scene)
    if args.spp:
        sc.c.samples_per_pixel = args.spp
    if args.width:
        sc.c.width = args.width
    if args.height:
        sc.c.height = args.height
    W, H, SPP, N_SPH = sc.c.width, sc.c.height, sc.c.samples_per_pixel, sc.c.n_spheres
    headline = args.scene == HEADLINE and not (args.spp or args.width or args.height or args.variant)
    # RCCL prints its banner through C stdio when the communicators come up: keep stdout for the ONE line
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        t_setup = time.perf_counter()
        grp = hip.HipGroup(sc.ptr, args.gpus)    # scene replicas + streams + threads + ncclCommInitAll: outside the timed region
        setup_s = time.perf_counter() - t_setup
    finally:
        _flush_c_stdio()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    if args.variant:
        grp.set_option("variant", args.variant)
    info = grp.info()
    devs = sorted(set(info["rank_devices"]))

    def sync_all():
        for d in devs:
            torch.cuda.synchronize(d)

    for _ in range(max(args.warmup, 2)):      # (two frames build the queue order, like the N = 1 path's warm-up)
        grp.render()
    sync_all()
    # ---- timed region: K frames PIPELINED two deep (rt_hip_group_submit / _collect): frame i's gather + de-interleave run
    # under frame i+1's kernels; every frame ends assembled in HBM of the first device.  `value` is this throughput.
    kern, gath = [], []
    t0 = time.perf_counter()
    grp.submit()
    for _ in range(args.steps - 1):
        grp.submit()
        st = grp.collect()
        kern.append(st["kernel_ms"]); gath.append(st["gather_ms"])
    st = grp.collect()
    kern.append(st["kernel_ms"]); gath.append(st["gather_ms"])
    sync_all()
    elapsed = time.perf_counter() - t0
    # ---- ONE frame at a time, nothing overlapped (the north star's case: G launches -> ONE gather -> de-interleave -> frame
    # on the first device): latency, and where its non-kernel time goes (RtStats.group_us: host clock since submit)
    lat, lkern, lgath, stages = [], [], [], []
    for _ in range(max(5, min(args.steps, 20))):
        st = grp.render()
        lat.append(st["frame_ms"]); lkern.append(st["kernel_ms"]); lgath.append(st["gather_ms"]); stages.append(st["group_us"])
    sync_all()
    # the assembled frame against ONE launch of the whole frame on the first device (same process): must be byte-identical
    frame, _ = grp.render_to_host()
    info = grp.info()                  # (again: a gather that failed to enqueue mid-run has switched the transport)
    ranks = grp.ranks()                # where every rank runs + its kernel of the last frame
    one = hip.HipScene(sc.ptr, devs[0])
    if args.variant:
        one.set_option("variant", args.variant)
    ks = []
    ref = None
    for _ in range(4):
        ref, st1 = one.render_to_host()
        ks.append(st1["kernel_ms"])
    one.close()
    n1_kernel_ms = sorted(ks[1:])[1]
    identical = bool(np.array_equal(frame, ref))
    grp.close()
    if not identical:
        raise BenchError(f"the {args.gpus}-rank frame differs from the single launch in {int((frame != ref).sum())} bytes")
    samples = W * H * SPP
    med = lambda v: sorted(v)[len(v) // 2]    # noqa: E731
    ms_per_step = elapsed * 1e3 / args.steps
    out = {
        "metric": "Msamples/sec (pixels x spp / s) on cover_scene 1200x800",
        "value": round(samples * args.steps / elapsed / 1e6, 3), "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "scene derived from the reference's data/cover_scene.json (committed under scenes/); Philox seed 0",
        "config": {"workload": f"{os.path.basename(args.scene)}: {W}x{H} spp {SPP} depth {sc.c.max_depth}, {N_SPH} spheres"
                               + (" (BASELINE configs[1])" if headline else " (NON-HEADLINE run)"),
                   "parallelism": f"single process, in-library group (rt_hip_group_*): {info['n_ranks']} ranks x interleaved {info['tile_rows']}-scanline tiles, "
                                  f"one host thread + render stream + transfer stream per rank, ONE {info['transport']} gather per frame, frames pipelined two deep "
                                  f"(rt_hip_group_submit / _collect; frame_latency_ms = one blocking frame)"
                                  + (" — RANKS SHARE DEVICES (RT_GPUS_EMULATE=1: a functional check, not a scaling number)" if info["emulated"] else ""),
                   "inputs": "scene tables resident in HBM of every device before the timed region; each timed frame ends assembled in HBM of the first device"},
        "value_note": "pipelined throughput (frame i's gather under frame i+1's kernels); frame_latency_ms is ONE blocking frame, speedup_vs_n1_latency the north star's figure",
        "kernel_ms": round(med(kern), 4),                 # slowest rank's shard, median over the timed (pipelined) frames
        "gather_ms": round(med(gath), 4),                 # device 0's clock: (rank 0's kernel start -> frame assembled) - slowest kernel
        "frame_latency_ms": round(med(lat), 4),           # ONE blocking frame: submit -> assembled in HBM of the first device
        "frame_latency_kernel_ms": round(med(lkern), 4),
        "frame_latency_non_kernel_ms": round(med([a - b for a, b in zip(lat, lkern)]), 4),
        "frame_latency_stages_us": {k: round(med([u[i] for u in stages]), 1) for i, k in enumerate(
            ("last_rank_thread_running", "last_rank_enqueued", "submitter_knows", "gather_enqueued", "submit_returns", "assembled_seen", "frame_done", "stats_read"))},
        "n1_kernel_ms": round(n1_kernel_ms, 4),           # the whole frame in one launch on the first device, same process
        "speedup_vs_n1_latency": round(n1_kernel_ms / med(lat), 3),
        "speedup_vs_n1_pipelined": round(n1_kernel_ms / ms_per_step, 3),
        "frame_identical_to_n1": identical,
        "rccl_ranks": info["rccl_comms"], "transport": info["transport"], "transport_fallback": info["transport_fallback"],
        "transport_fallback_reason": info["fallback_reason"] or None, "rank_devices": info["rank_devices"],
        "distinct_devices": info["n_devices"], "visible_gpus": visible, "setup_ms": round(setup_s * 1e3, 1),
        # one line explains its own efficiency: per rank, its kernel of the last blocking frame, when its host thread ran and had its
        # launch enqueued (us since submit), where its device sits and whether it reaches the first device directly
        "per_rank": {"kernel_ms": [round(r["kernel_ms"], 4) for r in ranks], "t_wake_us": [round(r["t_wake_us"], 1) for r in ranks],
                     "t_enq_us": [round(r["t_enq_us"], 1) for r in ranks], "pci_bus_id": [r["pci_bus_id"] for r in ranks],
                     "numa_node": [r["numa_node"] for r in ranks], "pinned_cpus": [r["pinned_cpus"] for r in ranks],
                     "peer_to_root": [r["peer_to_root"] for r in ranks]},
        "git_head": _git_head(),
    }
    # the N > 1 line parses like the N = 1 line: the roofline of the dominant kernel per rank, and the CPU baseline (a pointer)
    out["roofline"] = _group_roofline("cfg2", args.gpus, out["per_rank"]["kernel_ms"]) if headline else None
    out["cpu_baseline"] = _n1_cpu_baseline_pointer()
    if headline and not args.no_other_configs:
        try:   # (the extra must never cost the line)
            out["configs3_on_group"] = configs3_on_group_inlib(pkg, args.gpus, devs[0])
        except BaseException as e:   # noqa: BLE001
            if isinstance(e, KeyboardInterrupt):
                raise
            out["configs3_on_group"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return json.dumps(out)


def _init_collective(torch, dist, rank, world, dev):
    """Bring up the process groups of an N-rank run so that the TRANSPORT can never cost the run.  The default group is gloo
    (TCP on 127.0.0.1: the control plane — barriers, the ranks' statistics); the tiles travel on a second group, RCCL
    ("nccl"), IF it comes up on every rank and a first gather of known bytes arrives intact on rank 0 — decided by all ranks
    together over gloo, so nobody waits in a collective the others have given up on.  Otherwise (communicator error, a
    gather that raises, does not complete within 90 s (polled) or delivers wrong bytes; RT_BENCH_FORCE_TRANSPORT=gloo: tests):
    the tiles are staged through pinned host memory and gathered over gloo ("gloo-host"): the same frame, no xGMI, and the
    line says so.  Returns (transport, data group, note)."""
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world)
    forced = os.environ.get("RT_BENCH_FORCE_TRANSPORT", "")
    ok, why, pg = 1, "", None
    if forced == "gloo":
        ok, why = 0, "RT_BENCH_FORCE_TRANSPORT=gloo"
    else:
        # The self-test must not hang the run: its gather is issued asynchronously and POLLED against a deadline (a collective
        # that never completes leaves is_completed() False and the ranks agree on the fall-back over gloo).  The timed loop keeps
        # torch's default wait semantics — work.wait() enqueues a stream wait, the host runs on — which the two-deep pipeline of
        # FramePipeline relies on (TORCH_NCCL_BLOCKING_WAIT, set process-wide until round 5, would have made every wait a host block).
        try:
            if os.environ.get("RT_BENCH_INJECT_NCCL_FAILURE") == "1":
                raise RuntimeError("injected (RT_BENCH_INJECT_NCCL_FAILURE=1)")
            # (no watchdog abort: a collective that never completes must not take the process — and the JSON line — down; the
            #  self-test below has its own deadline, well inside the group's)
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
            pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=600))   # nccl == RCCL on ROCm; communicators come up with the first collective
            probe = torch.full((4096,), rank + 1, dtype=torch.uint8, device=dev)
            outs = [torch.zeros(4096, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
            work = dist.gather(probe, outs, dst=0, group=pg, async_op=True)
            deadline = time.perf_counter() + float(os.environ.get("RT_BENCH_SELFTEST_TIMEOUT_S", "90"))
            while not work.is_completed():
                if time.perf_counter() > deadline:
                    raise TimeoutError("self-test gather did not complete")
                time.sleep(0.002)
            work.wait()
            torch.cuda.synchronize()
            if rank == 0:
                got = [int(o[-1].item()) for o in outs]
                if got != list(range(1, world + 1)):
                    raise RuntimeError(f"self-test gather delivered {got}")
        except BaseException as e:   # noqa: BLE001
            if isinstance(e, KeyboardInterrupt):
                raise
            ok, why = 0, f"{type(e).__name__}: {e}"[:300]
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # (gloo, CPU tensor: every rank learns whether ALL ranks are fine)
    if int(flag[0]) == 1:
        return "rccl", pg, ""
    whys = [None] * world
    dist.all_gather_object(whys, why)
    if pg is not None:
        try:
            dist.destroy_process_group(pg)
        except Exception:
            pass
    note = "; ".join(f"rank {r}: {w}" for r, w in enumerate(whys) if w) or "another rank failed"
    return "gloo-host", None, note


def _device_place(torch, local_rank):
    """where this rank's GPU sits: PCI bus id, the NUMA node sysfs reports, peer access to device 0 of the node"""
    prop = torch.cuda.get_device_properties(local_rank)
    pci = ""
    for attr in ("pci_bus_id", "pci_device_id", "pci_domain_id"):
        if not hasattr(prop, attr):
            pci = ""
            break
    else:
        pci = f"{prop.pci_domain_id:04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
    numa = -1
    try:
        numa = int(open(f"/sys/bus/pci/devices/{pci}/numa_node").read())
    except Exception:
        pass
    peer = 1
    try:
        peer = int(torch.cuda.can_device_access_peer(local_rank, 0)) if local_rank != 0 else 1
    except Exception:
        peer = -1
    return {"pci_bus_id": pci, "numa_node": numa, "peer_to_root": peer, "uuid": str(getattr(prop, "uuid", "")), "host": os.uname().nodename, "local_rank": local_rank}


def _pin_to_device(place):
    """One process per GPU: keep this rank's host threads on the CPUs of its GPU's NUMA node (what rt_hip_group does for its
    rank threads inside the library; RT_GROUP_PIN=0: not).  Returns the number of CPUs pinned to, 0 = left alone."""
    if os.environ.get("RT_GROUP_PIN") == "0" or not place.get("pci_bus_id") or not hasattr(os, "sched_setaffinity"):
        return 0
    try:
        cpus = set()
        for part in open(f"/sys/bus/pci/devices/{place['pci_bus_id']}/local_cpulist").read().strip().split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return 0
        os.sched_setaffinity(0, cpus)
        return len(cpus)
    except Exception:
        return 0


def run_ranks(args):
    import torch
    import torch.distributed as dist

    os.chdir(ROOT)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise BenchError("bench.py needs a GPU: the hot path has no CPU fallback")
    n_visible = torch.cuda.device_count()
    if local_rank >= n_visible:   # (a launcher that narrows HIP_VISIBLE_DEVICES per rank leaves every rank ONE device, ordinal 0)
        if n_visible == 1:
            local_rank = 0
        else:
            raise BenchError(f"LOCAL_RANK {local_rank} but {n_visible} device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_COLLECTIVE=1: a one-rank RCCL group still goes through init / gather / barrier — the N > 1
    # code path exercised on a 1-GPU box (self-check; the line then says so in config.parallelism)
    force_coll = world == 1 and os.environ.get("RT_BENCH_FORCE_COLLECTIVE") == "1"
    collective = world > 1 or force_coll
    transport, data_group, transport_note = "none", None, ""
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner through C stdio when the communicator comes up (device_id = eager init); on
        # a pipe it would sit in the buffer until exit and land AFTER the JSON line.  stdout carries that ONE line
        # only: while the group comes up, file descriptor 1 points at stderr, and the buffer is pushed out there.
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            transport, data_group, transport_note = _init_collective(torch, dist, rank, world, dev)
        finally:
            _flush_c_stdio()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    pkg = graft.load_package()
    abi, host, hip = pkg.abi, pkg.host, pkg.hip
    from rust_raytracer_amd import dist as rdist

    sc = host.Scene.load(args.scene)
    if args.spp:
        sc.c.samples_per_pixel = args.spp
    if args.width:
        sc.c.width = args.width
    if args.height:
        sc.c.height = args.height
    W, H, SPP, N_SPH = sc.c.width, sc.c.height, sc.c.samples_per_pixel, sc.c.n_spheres
    headline = args.scene == HEADLINE and not (args.spp or args.width or args.height or args.variant)

    gs = hip.HipScene(sc.ptr, local_rank)          # scene tables + textures -> HBM (outside the timed region)
    if args.variant:
        gs.set_option("variant", args.variant)
    tiles = rdist.shard(rank, world)
    pipe = rdist.FramePipeline(H, W, rank, world, dev, force_collective=force_coll,   # double-buffered tiles; frame i's gather runs under frame i+1
                               group=data_group if collective else None, host_staged=collective and transport == "gloo-host")
    stream = torch.cuda.current_stream()

    # every rank renders on its own GPU: collect (host, PCI bus id / uuid) of each rank's device and compare
    ranks_devices = None
    place = _device_place(torch, local_rank)
    place["pinned_cpus"] = _pin_to_device(place) if world > 1 else 0
    if collective:
        gathered = [None] * world
        dist.all_gather_object(gathered, place)
        ranks_devices = gathered
        shared = len({(g["host"], g["uuid"], g["pci_bus_id"], g["local_rank"]) for g in gathered}) != world
        # (RT_BENCH_ALLOW_SHARED_DEVICE=1: a functional run of the process-per-GPU path with the ranks on ONE GPU — tests; the line says so)
        if shared and os.environ.get("RT_BENCH_ALLOW_SHARED_DEVICE") != "1":
            raise BenchError(f"ranks share a device: {gathered}")

    def fence():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    n_frames = 0
    for i in range(args.warmup):
        buf, _ = pipe.begin(i)
        gs.render(buf.data_ptr(), 0, tiles, stream.cuda_stream)   # the megakernel, on torch's current stream
        pipe.submit(i)
    pipe.drain()
    fence()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        buf, done = pipe.begin(i)            # (hands back frame i-2, assembled on rank 0)
        n_frames += done is not None
        ev0[i].record(stream)
        gs.render(buf.data_ptr(), 0, tiles, stream.cuda_stream)
        ev1[i].record(stream)
        pipe.submit(i)                       # N > 1: ONE gather over RCCL, asynchronous
    n_frames += sum(f is not None for f in pipe.drain())
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / max(1, args.steps)
    st = gs.wait()                                  # counters of the last launch (this rank's shard)

    # ---- ONE frame, nothing overlapped (the north star's case: render -> one gather at frame end -> frame on rank 0)
    lat_ms = None
    if collective:
        lats = []
        for i in range(7):
            fence()
            l0 = time.perf_counter()
            buf, _ = pipe.begin(i)
            gs.render(buf.data_ptr(), 0, tiles, stream.cuda_stream)
            pipe.submit(i)
            frames = pipe.drain()              # waits for the gather, permutes the rows on rank 0
            torch.cuda.synchronize()
            if collective:
                dist.barrier()                 # the frame is complete everywhere (rank 0 holds it)
            lats.append((time.perf_counter() - l0) * 1e3)
            assert (frames[0] is not None) == (rank == 0)
        lat_ms = sorted(lats[2:])[len(lats[2:]) // 2]   # median of 5 after 2 warm-ups
        gs.wait()
    # the whole frame on ONE GPU (rank 0), for the N = 1 reference inside the same line
    n1_kernel_ms = None
    if world > 1:
        if rank == 0:
            full = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
            ks = []
            for _ in range(4):
                gs.render(full.data_ptr(), 0, None, stream.cuda_stream)
                ks.append(gs.wait()["kernel_ms"])
            n1_kernel_ms = sorted(ks[1:])[1]
        dist.barrier()

    # ---- BASELINE configs[3] (the workload BASELINE.json names for the 8-GPU node) on the SAME ranks: 1 warm + 3 blocking frames
    c3 = None
    if world > 1 and headline and not args.no_other_configs:
        try:
            c3 = _configs3_on_ranks(torch, dist, pkg, rdist, rank, world, local_rank, dev, stream, data_group, transport)
        except BaseException as e:   # noqa: BLE001 — (every rank raises or none: the calls inside are collective)
            if isinstance(e, KeyboardInterrupt):
                raise
            c3 = {"error": f"{type(e).__name__}: {e}"[:300]}

    t = torch.tensor([elapsed, kernel_ms, float(st["segments"]), float(st["exact_tests"]), float(st["grid_steps"]), lat_ms or 0.0],
                     dtype=torch.float64)   # (CPU: the default group is gloo)
    per_rank = None
    if collective:   # one line explains its own efficiency: every rank's kernel time (mean over the timed frames) and wall time
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"kernel_ms": round(kernel_ms, 4), "elapsed_ms": round(elapsed * 1e3, 3), "samples_share": round(st["samples"] / max(1, W * H * SPP), 5)})
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms, lat_ms = float(tmax[0]), float(tmax[1]), float(tmax[5])
        segments, exact, steps = float(tsum[2]), float(tsum[3]), float(tsum[4])
    else:
        segments, exact, steps = float(st["segments"]), float(st["exact_tests"]), float(st["grid_steps"])

    if rank == 0:
        assert n_frames == args.steps, (n_frames, args.steps)   # every timed frame was assembled inside the timed region
        samples = W * H * SPP
        ms_per_step = elapsed * 1e3 / args.steps
        value = samples * args.steps / elapsed / 1e6
        kernel_s = kernel_ms * 1e-3
        tests_per_launch = segments * N_SPH / world     # algorithmic tests of one launch (= one rank's shard)
        exact_per_launch = exact / world
        # ---- roofline of the dominant kernel (rt_megakernel): executed basis, counters from profiles/
        pmc_path, pmc = _latest_pmc()
        roof = {"bound": "valu", "kernel": "rt_megakernel", "unit": "T lane-slots/s", "peak": round(PEAK_LANE_SLOTS_T, 2),
                "achieved": None, "frac": None, "traffic": None,
                "note": "vector-ALU issue binds this path (no MFMA: no dense contraction; HBM traffic ~0.01 % of 8 TB/s x frame "
                        "time): frac = SQ_THREAD_CYCLES_VALU / (16 lanes x 1024 SIMDs x 2.4 GHz x kernel time) = VALU issue busy x "
                        "lane utilisation; counters per launch from rocprofv3 --pmc of the same build, time from this run's HIP events"}
        if pmc is not None and headline and world == 1:
            m = pmc["mean_per_launch"]
            thr, act = m["SQ_THREAD_CYCLES_VALU"], m["SQ_ACTIVE_INST_VALU"]
            roof["achieved"] = round(thr / kernel_s / 1e12, 3)
            roof["frac"] = round(thr / kernel_s / 1e12 / PEAK_LANE_SLOTS_T, 4)
            roof["valu_issue_busy"] = round(act / (N_SIMD * kernel_s * ENGINE_HZ / 4.0), 4)
            roof["lane_utilisation"] = round(thr / (64.0 * act), 4)
            roof["counters"] = {"SQ_THREAD_CYCLES_VALU": thr, "SQ_ACTIVE_INST_VALU": act, "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU"),
                                "source": os.path.relpath(pmc_path, ROOT), "kernel_ms_of_that_run": pmc.get("kernel_ms"),
                                "git_head_of_that_build": pmc.get("git_head"), "git_head_of_this_build": _git_head(),
                                "same_kernel_sources": (pmc.get("kernel_src_hash") == _build_info("kernel_src_hash")) if pmc.get("kernel_src_hash") else None}
            # the executed basis falls when work is REMOVED: the same frame priced with the lane work of the end-of-round-2
            # kernel (same scene, same image) says what the time alone bought
            try:
                r2 = json.load(open(os.path.join(ROOT, "profiles", "r02_run29_pmc.json")))["mean_per_launch"]["SQ_THREAD_CYCLES_VALU"]
                roof["work_removed_since_round2"] = {
                    "lane_cycles_round2_kernel": r2, "lane_cycles_this_kernel": thr, "removed_frac": round(1.0 - thr / r2, 4),
                    "frac_if_priced_with_round2_work": round(r2 / kernel_s / 1e12 / PEAK_LANE_SLOTS_T, 4),
                    "note": "frac fell from 0.50 because executed lane work fell faster than time; this is round 2's lane work / this run's time"}
            except Exception:
                pass
            if pmc.get("hbm_bytes_per_launch") is not None:
                alg_bytes = 3 * W * H + gs.query("table_bytes") + gs.query("texel_bytes")   # framebuffer + one pass over geometry / material cores / cell table / item lists (+ texels)
                roof["traffic"] = pmc["hbm_bytes_per_launch"]
                roof["traffic_source"] = pmc.get("source")
                roof["algorithmic_bytes"] = alg_bytes
                roof["traffic_over_algorithmic"] = round(pmc["hbm_bytes_per_launch"] / alg_bytes, 2)
        ex_tflops = exact_per_launch * FLOP_PER_TEST / kernel_s / 1e12
        roof["executed_f64"] = {"exact_tests_per_launch": int(exact_per_launch), "flop_per_test": FLOP_PER_TEST, "tflops": round(ex_tflops, 3),
                                "peak_fp64_vector_tflops": PEAK_FP64_VALU_TFLOPS, "frac": round(ex_tflops / PEAK_FP64_VALU_TFLOPS, 4)}
        alg_tflops = tests_per_launch * FLOP_PER_TEST / kernel_s / 1e12
        roof["algorithmic"] = {"tests_per_launch": int(tests_per_launch), "tflops_equivalent": round(alg_tflops, 2),
                               "algorithmic_speedup": round(tests_per_launch / max(1.0, exact_per_launch), 1),
                               "note": "the reference's brute force (segments x n_spheres, SURVEY §8d) over this kernel's time: a speed-up "
                                       "figure, not a roofline fraction — the grid walk executes ~1 % of these tests"}
        out = {
            "metric": "Msamples/sec (pixels x spp / s) on cover_scene 1200x800",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "scene derived from the reference's data/cover_scene.json (committed under scenes/); Philox seed 0",
            "config": {"workload": f"{os.path.basename(args.scene)}: {W}x{H} spp {SPP} depth {sc.c.max_depth}, {N_SPH} spheres"
                                   + (" (BASELINE configs[1])" if headline else " (NON-HEADLINE run)"),
                       "parallelism": (f"one process per GPU (torch.distributed.run): {world} x interleaved {rdist.TILE_ROWS}-scanline tiles, one "
                                       + ("RCCL gather per frame through torch.distributed (overlapping the next frame's render)" if transport == "rccl" else
                                          "gather per frame over gloo with the tiles staged through pinned host memory — RCCL DID NOT COME UP, see transport_fallback_reason")) if world > 1
                                      else (f"single GPU (one-rank {'RCCL' if transport == 'rccl' else 'gloo, host-staged'} group forced: self-check)" if force_coll else "single GPU"),
                       "inputs": "scene tables resident in HBM before the timed region"},
            "kernel_ms": round(kernel_ms, 4), "segments_per_sample": round(segments / samples, 4),
            "exact_tests_per_segment": round(exact / max(1.0, segments), 3),
            "grid_steps_per_segment": round(steps / max(1.0, segments), 3),
            "roofline": roof,
        }
        if collective:
            out["rccl_ranks"] = dist.get_world_size() if transport == "rccl" else 0
            out["visible_gpus"] = torch.cuda.device_count()
            out["rank_devices"] = [f"{g['host']}:{g['pci_bus_id'] or g['local_rank']}" for g in ranks_devices]
            out["ranks_share_devices"] = len(set(out["rank_devices"])) != len(out["rank_devices"])   # (True only under RT_BENCH_ALLOW_SHARED_DEVICE=1: a functional check, not a scaling number)
            out["transport"] = transport
            out["transport_fallback"] = transport != "rccl"
            out["transport_fallback_reason"] = transport_note or None
            out["per_rank"] = {"kernel_ms": [p["kernel_ms"] for p in per_rank], "elapsed_ms": [p["elapsed_ms"] for p in per_rank],
                               "samples_share": [p["samples_share"] for p in per_rank], "pci_bus_id": [g["pci_bus_id"] for g in ranks_devices],
                               "numa_node": [g["numa_node"] for g in ranks_devices], "pinned_cpus": [g.get("pinned_cpus", 0) for g in ranks_devices],
                               "peer_to_root": [g["peer_to_root"] for g in ranks_devices]}
            out["frame_latency_ms"] = round(lat_ms, 4)      # ONE frame: render (slowest rank) + gather + row permutation, no overlap
            out["frame_latency_msamples_per_s"] = round(samples / lat_ms / 1e3, 1)
            if n1_kernel_ms is not None:
                out["n1_kernel_ms"] = round(n1_kernel_ms, 4)                      # the whole frame on rank 0's GPU alone, same process
                out["speedup_vs_n1_latency"] = round(n1_kernel_ms / lat_ms, 3)    # the north star's ">= 6x at 8 GPUs" figure
                out["speedup_vs_n1_pipelined"] = round(n1_kernel_ms / ms_per_step, 3)
            if world > 1:   # the N > 1 line parses like the N = 1 line
                if headline:
                    out["roofline"] = _group_roofline("cfg2", world, [p["kernel_ms"] for p in per_rank])
                out["cpu_baseline"] = _n1_cpu_baseline_pointer()
                if c3 is not None:
                    out["configs3_on_group"] = c3
        if world == 1:
            # the frame as a host caller sees it with the scene resident — kernel + the 2.88 MB
            # device-to-host copy of the RGB8 frame (pageable numpy buffer) — reported beside `value`, never as it
            gs.render_to_host()
            h0 = time.perf_counter()
            for _ in range(3):
                gs.render_to_host()
            out["frame_ms_to_host_buffer"] = round((time.perf_counter() - h0) * 1e3 / 3, 3)
            # `value` is the steady state of repeated frames of one view (the queue order learned from the previous frame,
            # DESIGN.md §4.1); a one-shot render (rt_render_rgb8, the reference's one frame per process) has no previous
            # frame: the same kernel with the fixed bottom-row-first order — the FIRST frame of a fresh scene, best of 3
            fb1 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
            k1 = []
            for _ in range(4):
                g1 = hip.HipScene(sc.ptr, local_rank)
                g1.set_option("tile_order", 1)
                if args.variant:
                    g1.set_option("variant", args.variant)
                g1.render(fb1.data_ptr(), 0, None, stream.cuda_stream)
                k1.append(g1.wait()["kernel_ms"])
                g1.close()
            out["first_frame_kernel_ms"] = round(min(k1[1:]), 4)
            out["value_note"] = ("steady state: frame i uses the tile-queue order learned from frame i-1 of the same view; first_frame_kernel_ms = a scene's "
                                 "FIRST frame (what a one-shot rt_render_rgb8 gets): bottom row first; the measured order knows which tiles hold this seed's rare 50-segment paths, which no seed predicts (DESIGN.md §4.1)")
            out["git_head"] = _git_head()
        if world == 1 and headline and not args.no_other_configs:
            # (the extras must never cost the line: whatever goes wrong in one of them is reported in its place)
            def extra(fn, *a):
                try:
                    return fn(*a)
                except BaseException as e:   # noqa: BLE001
                    if isinstance(e, KeyboardInterrupt):
                        raise
                    return {"error": f"{type(e).__name__}: {e}"[:300]}
            out["other_configs"] = extra(other_configs, pkg, torch, dev, stream)
            out["mixed_radius_worlds"] = extra(mixed_radius_worlds, pkg, torch, dev, stream)
            out["cli"] = extra(cli_wall_times)
            out["animation"] = extra(animation_runs, kernel_ms, out["first_frame_kernel_ms"])
            out["cli_note"] = ("median of 3 fresh processes each; the HIP runtime's start-up in a fresh process (hip_init_ms) varies 50 - 230 ms from run to run "
                               "on these boxes and is most of the spread of wall_ms (profiles/r05_run14_cli_wall.log: 150 - 160 ms wall is the typical figure)")
        if world == 1 and not args.no_cpu_baseline:
            oracle = graft.load_oracle()
            cores = oracle.lib(abi).rt_oracle_threads()

            def cpu_run(stride, want_linear=False):
                ct = abi.RtRowTiles(1, 0, stride)
                c0 = time.perf_counter()
                o_rgb, o_lin, ost = oracle.render(abi, sc.ptr, ct, 0, want_linear=want_linear)
                return time.perf_counter() - c0, ost, abi.tiles_local_rows(H, ct), o_rgb, o_lin

            stride = max(1, args.cpu_row_stride)
            if stride == 0 or args.cpu_row_stride == 16:   # default: size the sample for <= ~25 s of wall time on this box
                probe_s, _, _, _, _ = cpu_run(64)          # (the GPU box's 128 host threads take the WHOLE frame in ~21 s)
                stride = int(min(64, max(1, round(64 * probe_s / 25.0))))
            csec, ost, rows, o_rgb, o_lin = cpu_run(stride, want_linear=True)
            out["cpu_baseline"] = {"value": round(ost["samples"] / csec / 1e6, 4), "unit": "Msamples/s",
                                   "cores": cores, "kind": "port",
                                   "sample": f"{'every scanline' if stride == 1 else ('every 2nd scanline' if stride == 2 else ('every 3rd scanline' if stride == 3 else f'every {stride}th scanline'))} of the same frame ({rows} rows, {ost['samples'] / 1e6:.2f} Msamples, "
                                             f"{csec:.1f} s); C oracle, OpenMP over 32-pixel blocks of a scanline, -O3 -march=x86-64-v3 -ffp-contract=off",
                                   "gpu_over_cpu": round(value / (ost["samples"] / csec / 1e6), 1)}
            # the image the oracle just produced is the parity check of THIS run's frame (the oracle as checker, after the
            # timed region): same bar as tests/parity.py
            import numpy as np
            fb = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
            fl = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            gs.render(fb.data_ptr(), fl.data_ptr(), None, stream.cuda_stream)
            gst = gs.wait()
            g_rgb, g_lin = fb.cpu().numpy()[::stride], fl.cpu().numpy()[::stride]
            atol = 2e-6 + 3e-8 * SPP
            dlin = float(np.abs(g_lin.astype(np.float64) - o_lin.astype(np.float64)).max())
            d8 = np.abs(g_rgb.astype(np.int16) - o_rgb.astype(np.int16))
            flips = int((d8 != 0).sum())
            ok = bool(dlin <= atol and int(d8.max()) <= 1 and flips <= max(2, int(5e-4 * d8.size)))
            out["parity_vs_oracle"] = {"rows": rows, "of": H, "whole_frame": stride == 1, "max_dlin": dlin, "atol": atol, "rgb8_flips": flips,
                                       "rgb8_values": int(d8.size), "max_rgb8_diff": int(d8.max()), "ok": ok,
                                       "segments_equal": (gst["segments"] == ost["segments"] - ost["segments_discarded"]) if stride == 1 else None}
            if not ok:
                raise BenchError(f"frame differs from the oracle: {out['parity_vs_oracle']}")
        line = json.dumps(out)
    gs.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    return line if rank == 0 else None


def other_configs(pkg, torch, dev, stream):
    """the other BASELINE configs at FULL size on this GPU, 1 warm-up + 2 timed frames each — frames under 50 ms (the
    reference's 1 ms test_scene): 5 warm-up + 20 timed frames, median — (parity of each is tested at full size in
    tests/test_gpu_parity.py): kernel time from HIP events, Msamples/s"""
    sys.path.insert(0, os.path.join(ROOT, "scenes"))
    import procedural
    res = []
    cases = [("configs[0] test_scene 800x600 spp16 depth8 (lights, textures, hollow glass)", "scenes/cfg1_test_800x600_spp16.json", None, "cfg1"),
             ("configs[2] cover 3840x2160 spp1024, earth/moon + sky textures", "scenes/cfg3_cover_4k_textured.json", None, "cfg3"),
             ("configs[3] cover 3840x2160 spp512 textured, on ONE GPU (the 8-GPU config)", "scenes/cfg4_cover_4k_textured_spp512.json", None, "cfg4"),
             ("configs[4] procedural 10 001 spheres 3840x2160 spp2048", None, dict(width=3840, height=2160, spp=2048, half=50, seed=0), "cfg5")]
    for name, path, proc, key in cases:
        s = pkg.host.Scene.load(path) if path else pkg.host.Scene.loads(procedural.make_json(**proc))
        g = pkg.hip.HipScene(s.ptr, dev.index or 0)
        fb = torch.zeros((s.c.height, s.c.width, 3), dtype=torch.uint8, device=dev)
        ks = []
        n_timed = 2
        for i in range(64):
            g.render(fb.data_ptr(), 0, None, stream.cuda_stream)
            stt = g.wait()
            ks.append(stt["kernel_ms"])
            if i == 0 and ks[0] < 50.0:
                n_timed = 24   # a millisecond frame right after an idle GPU reads the clock ramp, not the kernel: more frames, median
            if i >= n_timed:
                break
        k = sum(ks[1:]) / 2.0 if n_timed == 2 else sorted(ks[5:])[(len(ks) - 5) // 2]
        n = s.c.width * s.c.height * s.c.samples_per_pixel
        rec = {"config": name, "kernel_ms": round(k, 3), "msamples_per_s": round(n / k / 1e3, 1), "n_spheres": s.c.n_spheres,
               "segments_per_sample": round(stt["segments"] / n, 3), "exact_tests_per_segment": round(stt["exact_tests"] / max(1, stt["segments"]), 2)}
        # counters of this config (rocprofv3 --pmc passes, tools/pmc_scene.sh -> profiles/rNN_pmc_<key>.json), per launch
        pm = _scene_pmc(key)
        if pm is not None:
            m = pm[1].get("mean_per_launch", {})
            if "SQ_THREAD_CYCLES_VALU" in m:
                # the counters' lane cycles over BOTH clocks: the counter passes' own kernel time (counters on: up to 19 % slower
                # on the 1 ms test scene) and this run's — the first is the fraction that was measured, the second an upper bound
                thr = m["SQ_THREAD_CYCLES_VALU"]
                if pm[1].get("kernel_ms"):
                    rec["lane_slot_frac_on_counter_clock"] = round(thr / (pm[1]["kernel_ms"] * 1e-3) / 1e12 / PEAK_LANE_SLOTS_T, 4)
                    rec["counter_run_kernel_ms"] = pm[1]["kernel_ms"]
                rec["lane_slot_frac_on_this_runs_clock"] = round(thr / (k * 1e-3) / 1e12 / PEAK_LANE_SLOTS_T, 4)
                rec["lane_slot_frac"] = rec.get("lane_slot_frac_on_counter_clock", rec["lane_slot_frac_on_this_runs_clock"])
                rec["same_kernel_sources"] = (pm[1].get("kernel_src_hash") == _build_info("kernel_src_hash")) if pm[1].get("kernel_src_hash") else None
            if pm[1].get("hbm_bytes_per_launch") is not None:
                # FETCH_SIZE / WRITE_SIZE count the L2's fabric-side requests: L2 MISSES, whether the Infinity Cache (256 MiB: every
                # texture of every BASELINE config fits) or HBM serves them — not HBM bytes (MI355X_MICROARCH.md, HBM section)
                rec["l2_miss_bytes"] = pm[1]["hbm_bytes_per_launch"]
                rec["hbm_bytes_compulsory"] = 3 * s.c.width * s.c.height + g.query("table_bytes") + g.query("texel_bytes")
                rec["l2_miss_note"] = "fabric-side L2 requests (FETCH_SIZE x 2 + WRITE_SIZE); the working set (texels + framebuffer) fits the 256 MiB Infinity Cache, so HBM sees about hbm_bytes_compulsory"
            if pm[1].get("l2_hit_rate") is not None:
                rec["l2_hit_rate"] = pm[1]["l2_hit_rate"]
            rec["counters_source"] = os.path.relpath(pm[0], ROOT)
        res.append(rec)
        g.close()
        del fb
    return res


def mixed_radius_worlds(pkg, torch, dev, stream):
    """SURVEY §8 f2 outside BASELINE's sphere distribution: 10^4 spheres with radii over two decades (scenes/procedural.py
    radii="loguniform" / "bimodal"; parity: tests/test_gpu_parity.py::test_mixed_radius_10k_sphere_worlds) at 1920x1080 spp 128:
    what the single-level grid + `large` list make of them"""
    sys.path.insert(0, os.path.join(ROOT, "scenes"))
    import procedural
    res = []
    for radii in ("loguniform", "bimodal"):
        s = pkg.host.Scene.loads(procedural.make_json(width=1920, height=1080, spp=128, half=50, seed=0, radii=radii))
        g = pkg.hip.HipScene(s.ptr, dev.index or 0)
        fb = torch.zeros((s.c.height, s.c.width, 3), dtype=torch.uint8, device=dev)
        ks = []
        for _ in range(3):
            g.render(fb.data_ptr(), 0, None, stream.cuda_stream)
            st = g.wait()
            ks.append(st["kernel_ms"])
        k = min(ks[1:])
        n = s.c.width * s.c.height * s.c.samples_per_pixel
        res.append({"world": f"procedural, radii {radii}: {s.c.n_spheres} spheres, 1920x1080 spp 128 depth 50", "kernel_ms": round(k, 2),
                    "msamples_per_s": round(n / k / 1e3, 1), "segments_per_sample": round(st["segments"] / n, 3),
                    "exact_tests_per_segment": round(st["exact_tests"] / max(1, st["segments"]), 2), "grid_steps_per_segment": round(st["grid_steps"] / max(1, st["segments"]), 2),
                    "grid_cells": g.query("grid_cells"), "grid_items": g.query("grid_items"), "grid_large": g.query("grid_large"), "lds_tables": g.query("lds_tables")})
        g.close()
        del fb
    return res


def cli_wall_times():
    """What a drop-in user sees: wall time of `raytracer <scene> out.png` from process start to PNG on disk — the
    reference's one frame per process (main.rs:7-20) — for the headline config and the reference's test_scene (three JPEG
    decodes), split by the CLI's own clocks (RT_STATS=1): JSON, JPEG, HIP start-up (hip_init_ms: on a thread beside the load;
    hip_wait_ms: what of it the load did not cover), scene set-up (tables + upload + module load + warm-ups), frame (the window
    the reference times, raytracer.rs:259-263), PNG.  Median of 3 runs each."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "rust-raytracer_amd", "raytracer")
    res = []
    if not os.path.exists(exe):
        return res
    for name, scene in (("cfg2 cover 1200x800 spp128", HEADLINE), ("cfg1 test_scene 800x600 spp16 (3 JPEG textures)", "scenes/cfg1_test_800x600_spp16.json")):
        runs = []
        with tempfile.TemporaryDirectory() as td:
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([exe, scene, os.path.join(td, "out.png")], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, RT_STATS="1"), timeout=120)
                wall = (time.perf_counter() - t0) * 1e3
                if r.returncode != 0:
                    continue
                try:
                    st = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
                except Exception:
                    continue
                st["wall_ms"] = wall
                runs.append(st)
        if not runs:
            continue
        runs.sort(key=lambda d: d["wall_ms"])
        m = runs[len(runs) // 2]
        res.append({"scene": name, "wall_ms": round(m["wall_ms"], 1), "main_ms": round(m["main_ms"], 1),
                    "process_start_and_exit_ms": round(m["wall_ms"] - m["main_ms"], 1),
                    "load_ms": round(m["load_ms"], 2), "json_ms": round(m["json_ms"], 2), "jpeg_ms": round(m["jpeg_ms"], 2),
                    "hip_init_ms": round(m["hip_init_ms"], 1), "hip_wait_ms": round(m.get("hip_wait_ms", m["hip_init_ms"]), 1), "setup_ms": round(m["setup_ms"], 1), "frame_ms": round(m["frame_ms"], 2),
                    "setup_plus_frame_ms": round(m["setup_ms"] + m["frame_ms"], 2),   # (what compares with the reference's "Frame time": its window opens on a scene that is simply in memory)
                    "kernel_ms": round(m["kernel_ms"], 2), "png_ms": round(m["png_ms"], 2), "runs": len(runs),
                    "setup_profile_ms": {k: round(v, 2) for k, v in (m.get("setup_profile") or {}).items()}})
    return res


def animation_runs(steady_kernel_ms, first_frame_kernel_ms, frames=32, orbit=3.0):
    """SURVEY §8 f4 — the reference's headline workflow (README.md:43-57 `anim/frame_%03d.png`, main.rs:17): `raytracer <scene>
    <prefix> --frames 32 --orbit 3` in a FRESH process for the headline config and the reference's test scene: frames per second
    from the first submit to the last PNG on disk, the per-frame kernel times under a MOVING camera (the tile-queue order is
    learned per view: does a 3 degree step keep it?), the PNG writer's times, and what bounds the run:
    overlap_efficiency = frames/s x max(kernel, png) — 1.0 = the slower of the two stages is never idle."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "rust-raytracer_amd", "raytracer")
    res = []
    if not os.path.exists(exe):
        return res
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    for name, scene, steady, first in (("cfg2 cover 1200x800 spp128", HEADLINE, steady_kernel_ms, first_frame_kernel_ms),
                                       ("cfg1 test_scene 800x600 spp16 (lights, textures)", "scenes/cfg1_test_800x600_spp16.json", None, None)):
        with tempfile.TemporaryDirectory() as td:
            t0 = time.perf_counter()
            r = subprocess.run([exe, scene, os.path.join(td, "frame"), "--frames", str(frames), "--orbit", str(orbit)], capture_output=True, text=True,
                               cwd=ROOT, env=dict(os.environ, RT_STATS="1"), timeout=300)
            wall = time.perf_counter() - t0
            n_png = len([f for f in os.listdir(td) if f.endswith(".png")])
            png_bytes = sum(os.path.getsize(os.path.join(td, f)) for f in os.listdir(td))
        if r.returncode != 0:
            res.append({"scene": name, "error": r.stderr[-300:]})
            continue
        st = json.loads([l for l in r.stderr.splitlines() if l.startswith('{"animation"')][-1])
        k, pz = st["kernel_ms"], st["png_ms"]
        moving = k[2:] if len(k) > 4 else k          # (frames 0, 1: no previous frame's order yet)
        rec = {"scene": name, "command": f"raytracer {scene} <prefix> --frames {frames} --orbit {orbit:g}", "frames": st["frames"], "pngs_on_disk": n_png,
               "frames_per_s": round(st["frames_per_s"], 2), "ms_per_frame": round(1e3 / st["frames_per_s"], 3),
               "process_wall_s": round(wall, 3), "frames_per_s_incl_process_startup": round(st["frames"] / wall, 2), "setup_ms": round(st["setup_ms"], 1),
               "kernel_ms_moving_camera": {"median": round(med(moving), 3), "min": round(min(moving), 3), "max": round(max(moving), 3), "first_frame": round(k[0], 3)},
               "kernel_ms_series": [round(x, 2) for x in k],
               "kernel_ms_note": "frames of an animation OVERLAP (every other frame through a second view + stream fills the previous frame's tail): a frame's kernel_ms counts from the later of its own start and the previous frame's kernel end",
               "png_ms": {"median": round(med(pz), 2), "max": round(max(pz), 2)}, "png_writers": st.get("png_writers"), "png_mb_per_frame": round(png_bytes / max(1, n_png) / 1e6, 3),
               "host_us_per_frame": st.get("host_us_per_frame"),   # the submitting thread: waiting for a buffer / camera + submit (the FIRST submit allocates the pinned staging buffer and brings the device-to-host copy engine up: ~9 ms, 0.3 ms per frame of 32) / collect / stdout + hand-over
               "bound_by": "kernel" if med(moving) >= med(pz) / max(1, st.get("png_writers") or 1) else "png",
               "overlap_efficiency": round(st["frames_per_s"] * max(med(moving), med(pz) / max(1, st.get("png_writers") or 1)) / 1e3, 3),
               # what the run spent beside frames x the slower stage: pipeline fill / drain and the FIRST submit (pinned staging buffers + the
               # device-to-host copy engine's queue, ~9 ms once) — 1 % of a 32-frame cfg2 run, 30 % of 32 frames of the 0.9 ms test scene
               "transients_ms": round(st["wall_s"] * 1e3 - st["frames"] * max(med(moving), med(pz) / max(1, st.get("png_writers") or 1)), 1)}
        if steady:
            rec["steady_state_kernel_ms_view0"] = round(steady, 3)
            rec["first_frame_kernel_ms_view0_fixed_order"] = first
        # Is the queue order kept when the camera moves?  The orbit changes the VIEW (12.3 - 15.3 ms over 93 degrees), so a moving frame
        # is compared with the steady state of ITS OWN view: the same resident scene, camera of frame f, rendered three times (the third
        # uses the order learned from the second: what `value` measures) — in this process, through the C ABI.
        try:
            probe = _same_view_steady_state(scene, orbit, [f for f in (4, 10, 16, 22, 28) if f < len(k)])
            rec["same_views"] = {"frames": probe["frames"], "moving_kernel_ms": [round(k[f], 3) for f in probe["frames"]],
                                 "steady_kernel_ms": probe["steady_kernel_ms"], "first_frame_fixed_order_kernel_ms": probe["fixed_order_kernel_ms"],
                                 "moving_over_steady": [round(k[f] / s_, 4) for f, s_ in zip(probe["frames"], probe["steady_kernel_ms"])]}
            rec["moving_over_steady"] = round(med(rec["same_views"]["moving_over_steady"]), 4)
        except BaseException as e:   # noqa: BLE001
            if isinstance(e, KeyboardInterrupt):
                raise
            rec["same_views"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        res.append(rec)
    return res


def _orbit_camera(host, sc, deg):
    """the CLI's orbit_camera (main.cpp): look_from turned about vup around look_at by `deg`, then Camera::new (camera.rs:45-77)"""
    import ctypes as C
    import math
    L = host.lib()
    cam = (C.c_double * 11)()
    L.rt_scene_camera.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.rt_scene_camera.restype = None
    L.rt_scene_camera(sc._h, cam)
    lf, la, up = list(cam[0:3]), list(cam[3:6]), list(cam[6:9])
    kl = math.sqrt(sum(u * u for u in up))
    kk = [u / kl for u in up]
    th = deg * (3.14159265358979323846264338327950288 / 180.0)
    c, s_ = math.cos(th), math.sin(th)
    v = [lf[i] - la[i] for i in range(3)]
    kv = sum(kk[i] * v[i] for i in range(3))
    kx = [kk[1] * v[2] - kk[2] * v[1], kk[2] * v[0] - kk[0] * v[2], kk[0] * v[1] - kk[1] * v[0]]
    frm = [la[i] + v[i] * c + kx[i] * s_ + kk[i] * kv * (1.0 - c) for i in range(3)]
    out = (C.c_double * 13)()
    L.rt_camera_derive((C.c_double * 3)(*frm), (C.c_double * 3)(*la), (C.c_double * 3)(*up), cam[9], cam[10], out)
    return list(out[0:3]), list(out[3:6]), list(out[6:9]), list(out[9:12])


def _same_view_steady_state(scene, orbit, frames):
    pkg = graft.load_package()
    sc = pkg.host.Scene.load(scene)
    steady, fixed = [], []
    for f in frames:
        cam = _orbit_camera(pkg.host, sc, orbit * f)
        g = pkg.hip.HipScene(sc.ptr, 0)
        g.set_camera(*cam)
        ks = [g.render_to_host()[1]["kernel_ms"] for _ in range(4)]
        g.close()
        fixed.append(round(ks[0], 3))             # a fresh scene's first frame: bottom row first
        steady.append(round(min(ks[2:]), 3))      # order learned from the previous frame of the SAME view
    return {"frames": frames, "steady_kernel_ms": steady, "fixed_order_kernel_ms": fixed}


def _n1_cpu_baseline_pointer():
    """N > 1 lines: the CPU baseline of the newest committed N = 1 bench line (profiles/rNN_runM_bench.json) — a pointer, not a second
    20 s oracle run inside a scaling measurement"""
    def run_key(p):
        b = os.path.basename(p)
        run = re.search(r"run(\d+)", b)
        return (b.split("_")[0], int(run.group(1)) if run else -1)
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")), key=run_key, reverse=True):
        try:
            j = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            continue
        cb = j.get("cpu_baseline")
        if j.get("n_gpus") == 1 and isinstance(cb, dict) and cb.get("value"):
            return {"value": cb["value"], "unit": cb.get("unit", "Msamples/s"), "cores": cb.get("cores"), "kind": cb.get("kind", "port"),
                    "sample": cb.get("sample"), "source": os.path.relpath(p, ROOT),
                    "note": "pointer to the N = 1 line's measurement (same C oracle, that box's host cores); not re-timed inside the scaling run"}
    return None


def _group_roofline(key, n_ranks, rank_kernel_ms):
    """N > 1 lines: the executed-basis roofline per RANK — the counters of the whole frame's launch (profiles/, one GPU) split evenly
    over the ranks (interleaved 2-scanline tiles: within +-2 %), each over ITS kernel time of this run"""
    pm = _latest_pmc() if key == "cfg2" else (_scene_pmc(key) or (None, None))
    roof = {"bound": "valu", "kernel": "rt_megakernel", "unit": "T lane-slots/s", "peak": round(PEAK_LANE_SLOTS_T, 2), "achieved": None, "frac": None, "traffic": None,
            "note": "per rank: SQ_THREAD_CYCLES_VALU of the whole frame's launch (rocprofv3 --pmc, one GPU, profiles/) / n_ranks / that rank's kernel time of THIS run; "
                    "frac = the slowest rank's (the one the frame waits for); peak per GPU"}
    if pm is None or pm[1] is None or not rank_kernel_ms:
        return roof
    m = pm[1].get("mean_per_launch", {})
    thr = m.get("SQ_THREAD_CYCLES_VALU")
    if not thr:
        return roof
    per_rank = [thr / n_ranks / (k * 1e-3) / 1e12 / PEAK_LANE_SLOTS_T if k > 0 else None for k in rank_kernel_ms]
    slow = max(rank_kernel_ms)
    roof["achieved"] = round(thr / n_ranks / (slow * 1e-3) / 1e12, 3)
    roof["frac"] = round(thr / n_ranks / (slow * 1e-3) / 1e12 / PEAK_LANE_SLOTS_T, 4)
    roof["per_rank_frac"] = [round(x, 4) if x is not None else None for x in per_rank]
    roof["counters"] = {"SQ_THREAD_CYCLES_VALU_whole_frame": thr, "source": os.path.relpath(pm[0], ROOT), "kernel_ms_of_that_run": pm[1].get("kernel_ms"),
                        "same_kernel_sources": (pm[1].get("kernel_src_hash") == _build_info("kernel_src_hash")) if pm[1].get("kernel_src_hash") else None}
    if pm[1].get("hbm_bytes_per_launch") is not None:
        roof["traffic"] = pm[1]["hbm_bytes_per_launch"] / n_ranks
    return roof


def _configs3_on_ranks(torch, dist, pkg, rdist, rank, world, local_rank, dev, stream, data_group, transport):
    """process-per-GPU form of configs3_on_group_inlib: every rank renders its interleaved tiles of BASELINE configs[3], ONE gather
    per frame on the run's data group; frame time = barrier -> frame assembled on rank 0 -> barrier (max over ranks)."""
    sc = pkg.host.Scene.load(CONFIGS3)
    W, H, SPP = sc.c.width, sc.c.height, sc.c.samples_per_pixel
    gs = pkg.hip.HipScene(sc.ptr, local_rank)
    tiles = rdist.shard(rank, world)
    pipe = rdist.FramePipeline(H, W, rank, world, dev, group=data_group, host_staged=transport == "gloo-host")
    fr, kn = [], []
    frame0 = None
    for i in range(4):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        buf, _ = pipe.begin(i)
        gs.render(buf.data_ptr(), 0, tiles, stream.cuda_stream)
        pipe.submit(i)
        frames = pipe.drain()
        torch.cuda.synchronize()
        dist.barrier()
        fr.append((time.perf_counter() - t0) * 1e3)
        kn.append(gs.wait()["kernel_ms"])
        if rank == 0:
            frame0 = frames[0]
    my = {"kernel_ms": sorted(kn[1:])[1], "frame_ms": sorted(fr[1:])[1]}
    allr = [None] * world
    dist.all_gather_object(allr, my)
    n1, identical = None, None
    if rank == 0:
        full = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
        for _ in range(2):
            gs.render(full.data_ptr(), 0, None, stream.cuda_stream)
            n1 = gs.wait()["kernel_ms"]
        identical = bool(torch.equal(frame0.to(full.device), full))
    dist.barrier()
    gs.close()
    if rank != 0:
        return None
    frame_ms = max(a["frame_ms"] for a in allr)
    per_rank = [round(a["kernel_ms"], 3) for a in allr]
    rec = {"workload": f"{os.path.basename(CONFIGS3)}: {W}x{H} spp {SPP} depth {sc.c.max_depth}, {sc.c.n_spheres} spheres, earth/moon + sky textures (BASELINE configs[3])",
           "frames": "1 warm + 3 blocking (median)", "frame_ms": round(frame_ms, 3), "kernel_ms": max(per_rank), "msamples_per_s": round(W * H * SPP / frame_ms / 1e3, 1),
           "n1_kernel_ms": round(n1, 3), "speedup_vs_n1": round(n1 / frame_ms, 3), "per_rank": {"kernel_ms": per_rank}, "transport": transport,
           "frame_identical_to_n1": identical, "roofline": _group_roofline("cfg4", world, per_rank)}
    if not identical:
        rec["error"] = "the assembled frame differs from the single launch"
    return rec


CONFIGS3 = "scenes/cfg4_cover_4k_textured_spp512.json"   # BASELINE configs[3]: cover 3840x2160 spp 512 textured, row-tiled across the GPUs


def configs3_on_group_inlib(pkg, n_gpus, first_device):
    """BASELINE configs[3] — the workload BASELINE.json names for 8 GPUs — on an in-library group of the SAME ranks: 1 warm + 3
    blocking frames (submit -> assembled in HBM of the first device), the slowest rank's kernel, every rank's kernel, and the
    whole frame in ONE launch on the first device (second of two frames) for the speed-up."""
    hip, host = pkg.hip, pkg.host
    sc = host.Scene.load(CONFIGS3)
    W, H, SPP = sc.c.width, sc.c.height, sc.c.samples_per_pixel
    samples = W * H * SPP
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        grp = hip.HipGroup(sc.ptr, n_gpus)
    finally:
        _flush_c_stdio()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    grp.render()
    fr, kn, per_rank = [], [], None
    for _ in range(3):
        st = grp.render()
        fr.append(st["frame_ms"]); kn.append(st["kernel_ms"])
        per_rank = [round(r["kernel_ms"], 3) for r in grp.ranks()]
    info = grp.info()
    frame, _ = grp.render_to_host()
    grp.close()
    one = hip.HipScene(sc.ptr, first_device)
    one.render_to_host()
    ref, st1 = one.render_to_host()
    one.close()
    import numpy as np
    identical = bool(np.array_equal(frame, ref))
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    rec = {"workload": f"{os.path.basename(CONFIGS3)}: {W}x{H} spp {SPP} depth {sc.c.max_depth}, {sc.c.n_spheres} spheres, earth/moon + sky textures (BASELINE configs[3])",
           "frames": "1 warm + 3 blocking (median)", "frame_ms": round(med(fr), 3), "kernel_ms": round(med(kn), 3), "msamples_per_s": round(samples / med(fr) / 1e3, 1),
           "n1_kernel_ms": round(st1["kernel_ms"], 3), "speedup_vs_n1": round(st1["kernel_ms"] / med(fr), 3), "per_rank": {"kernel_ms": per_rank},
           "transport": info["transport"], "frame_identical_to_n1": identical,
           "roofline": _group_roofline("cfg4", n_gpus, per_rank)}
    if not identical:
        rec["error"] = f"the {n_gpus}-rank frame differs from the single launch in {int((frame != ref).sum())} bytes"
    return rec


if __name__ == "__main__":
    main()
