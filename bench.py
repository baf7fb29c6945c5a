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
speed-up of this frame at 8 GPUs).  In the timed region the tile buffers are double-buffered:
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
  other_configs the other BASELINE configs at full size on one GPU (2 frames each): kernel_ms, Msamples/s
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


def main():
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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    os.chdir(ROOT)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_COLLECTIVE=1: a one-rank RCCL group still goes through init / gather / barrier — the N > 1
    # code path exercised on a 1-GPU box (self-check; the line then says so in config.parallelism)
    force_coll = world == 1 and os.environ.get("RT_BENCH_FORCE_COLLECTIVE") == "1"
    collective = world > 1 or force_coll
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
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
            dist.barrier()
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
    pipe = rdist.FramePipeline(H, W, rank, world, dev, force_collective=force_coll)   # double-buffered tiles; frame i's gather runs under frame i+1
    stream = torch.cuda.current_stream()

    # every rank renders on its own GPU: collect (host, PCI bus id / uuid) of each rank's device and compare
    ranks_devices = None
    if collective:
        prop = torch.cuda.get_device_properties(local_rank)
        ident = (os.uname().nodename, str(getattr(prop, "uuid", "")), str(getattr(prop, "pci_bus_id", local_rank)), local_rank)
        gathered = [None] * world
        dist.all_gather_object(gathered, ident)
        ranks_devices = gathered
        assert len({(g[0], g[1], g[2]) for g in gathered}) == world, f"ranks share a device: {gathered}"

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

    t = torch.tensor([elapsed, kernel_ms, float(st["segments"]), float(st["exact_tests"]), float(st["grid_steps"]), lat_ms or 0.0],
                     dtype=torch.float64, device=dev)
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
                                "source": os.path.relpath(pmc_path, ROOT), "kernel_ms_of_that_run": pmc.get("kernel_ms")}
            if pmc.get("hbm_bytes_per_launch") is not None:
                alg_bytes = 3 * W * H + 32 * N_SPH * 2 + 8 * 4800   # framebuffer + one pass over geometry/material/cell tables
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
                       "parallelism": f"{world} x interleaved {rdist.TILE_ROWS}-scanline tiles, one RCCL gather per frame (overlapping the next frame's render)" if world > 1 else ("single GPU (one-rank RCCL group forced: self-check)" if force_coll else "single GPU"),
                       "inputs": "scene tables resident in HBM before the timed region"},
            "kernel_ms": round(kernel_ms, 4), "segments_per_sample": round(segments / samples, 4),
            "exact_tests_per_segment": round(exact / max(1.0, segments), 3),
            "grid_steps_per_segment": round(steps / max(1.0, segments), 3),
            "roofline": roof,
        }
        if collective:
            out["rccl_ranks"] = dist.get_world_size()
            out["visible_gpus"] = torch.cuda.device_count()
            out["rank_devices"] = [f"{g[0]}:{g[2]}" for g in ranks_devices]
            out["frame_latency_ms"] = round(lat_ms, 4)      # ONE frame: render (slowest rank) + gather + row permutation, no overlap
            out["frame_latency_msamples_per_s"] = round(samples / lat_ms / 1e3, 1)
            if n1_kernel_ms is not None:
                out["n1_kernel_ms"] = round(n1_kernel_ms, 4)                      # the whole frame on rank 0's GPU alone, same process
                out["speedup_vs_n1_latency"] = round(n1_kernel_ms / lat_ms, 3)    # the north star's ">= 6x at 8 GPUs" figure
                out["speedup_vs_n1_pipelined"] = round(n1_kernel_ms / ms_per_step, 3)
        if world == 1:
            # the frame as a host caller sees it with the scene resident — kernel + the 2.88 MB
            # device-to-host copy of the RGB8 frame (pageable numpy buffer) — reported beside `value`, never as it
            gs.render_to_host()
            h0 = time.perf_counter()
            for _ in range(3):
                gs.render_to_host()
            out["frame_ms_to_host_buffer"] = round((time.perf_counter() - h0) * 1e3 / 3, 3)
        if world == 1 and headline and not args.no_other_configs:
            out["other_configs"] = other_configs(pkg, torch, dev, stream)
        if world == 1 and not args.no_cpu_baseline:
            oracle = graft.load_oracle()
            cores = oracle.lib(abi).rt_oracle_threads()

            def cpu_run(stride):
                ct = abi.RtRowTiles(1, 0, stride)
                c0 = time.perf_counter()
                _, _, ost = oracle.render(abi, sc.ptr, ct, 0, want_linear=False)
                return time.perf_counter() - c0, ost, abi.tiles_local_rows(H, ct)

            stride = max(1, args.cpu_row_stride)
            if stride == 0 or args.cpu_row_stride == 16:   # default: size the sample for ~10 s of wall time on this box
                probe_s, _, _ = cpu_run(64)
                stride = int(min(64, max(1, round(64 * probe_s / 10.0))))
            csec, ost, rows = cpu_run(stride)
            out["cpu_baseline"] = {"value": round(ost["samples"] / csec / 1e6, 4), "unit": "Msamples/s",
                                   "cores": cores, "kind": "port",
                                   "sample": f"{'every scanline' if stride == 1 else ('every 2nd scanline' if stride == 2 else ('every 3rd scanline' if stride == 3 else f'every {stride}th scanline'))} of the same frame ({rows} rows, {ost['samples'] / 1e6:.2f} Msamples, "
                                             f"{csec:.1f} s); C oracle, OpenMP over 32-pixel blocks of a scanline, -O3 -march=native -ffp-contract=off",
                                   "gpu_over_cpu": round(value / (ost["samples"] / csec / 1e6), 1)}
        line = json.dumps(out)
    gs.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        print(line, flush=True)   # the ONE JSON line, and the last thing this process writes to stdout


def other_configs(pkg, torch, dev, stream):
    """the other BASELINE configs at FULL size on this GPU, 1 warm-up + 2 timed frames each (parity of each is
    tested at full size in tests/test_gpu_parity.py): kernel time from HIP events, Msamples/s"""
    sys.path.insert(0, os.path.join(ROOT, "scenes"))
    import procedural
    res = []
    cases = [("configs[0] test_scene 800x600 spp16 depth8 (lights, textures, hollow glass)", "scenes/cfg1_test_800x600_spp16.json", None),
             ("configs[2] cover 3840x2160 spp1024, earth/moon + sky textures", "scenes/cfg3_cover_4k_textured.json", None),
             ("configs[3] cover 3840x2160 spp512 textured, on ONE GPU (the 8-GPU config)", "scenes/cfg4_cover_4k_textured_spp512.json", None),
             ("configs[4] procedural 10 001 spheres 3840x2160 spp2048 (tables in L2)", None, dict(width=3840, height=2160, spp=2048, half=50, seed=0))]
    for name, path, proc in cases:
        s = pkg.host.Scene.load(path) if path else pkg.host.Scene.loads(procedural.make_json(**proc))
        g = pkg.hip.HipScene(s.ptr, dev.index or 0)
        fb = torch.zeros((s.c.height, s.c.width, 3), dtype=torch.uint8, device=dev)
        ks = []
        for _ in range(3):
            g.render(fb.data_ptr(), 0, None, stream.cuda_stream)
            stt = g.wait()
            ks.append(stt["kernel_ms"])
        k = sum(ks[1:]) / 2.0
        n = s.c.width * s.c.height * s.c.samples_per_pixel
        res.append({"config": name, "kernel_ms": round(k, 3), "msamples_per_s": round(n / k / 1e3, 1), "n_spheres": s.c.n_spheres,
                    "segments_per_sample": round(stt["segments"] / n, 3), "exact_tests_per_segment": round(stt["exact_tests"] / max(1, stt["segments"]), 2)})
        g.close()
        del fb
    return res


if __name__ == "__main__":
    main()
