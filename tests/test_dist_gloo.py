"""N>1 path on CPU: two processes (gloo), each renders its interleaved scanline tiles, ONE
gather assembles the frame on rank 0 (rust-raytracer_amd/dist.py).  The row renderer here
is the oracle (tests may use it); on the GPU box bench.py plugs the HIP megakernel into the
same shard/gather code with backend nccl (= RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    os.chdir(ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = graft.load_package()
        oracle = graft.load_oracle()
        from rust_raytracer_amd import dist as rdist
        sc = pkg.host.Scene.load("scenes/cfg2_cover_1200x800_spp128.json")
        sc.c.width, sc.c.height, sc.c.samples_per_pixel = w, h, 2
        tiles = rdist.shard(rank, world)
        rgb, _, _ = oracle.render(pkg.abi, sc.ptr, tiles, n_threads=2, want_linear=False)
        pad = rdist.max_local_rows(h, world)
        local = torch.zeros((pad, w, 3), dtype=torch.uint8)
        local[: rgb.shape[0]] = torch.from_numpy(rgb)
        frame = rdist.gather_frame(local, h, w, rank, world)
        if rank == 0:
            np.save(out_path, frame.numpy())
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(40, 30), (33, 8), (16, 5)])  # ragged: last tile short / fewer tiles than ranks
def test_two_rank_gather_matches_single(tmp_path, oracle, abi, load_scene, w, h):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, out), nprocs=2, join=True)
    sc = load_scene("cover", w, h, 2)
    full, _, _ = oracle.render(abi, sc.ptr, want_linear=False)
    assert np.array_equal(np.load(out), full)


def _pipeline_worker(rank, world, port, w, h, n_frames, out_path, host_staged=False):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    os.chdir(ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = graft.load_package()
        oracle = graft.load_oracle()
        from rust_raytracer_amd import dist as rdist
        sc = pkg.host.Scene.load("scenes/cfg2_cover_1200x800_spp128.json")
        sc.c.width, sc.c.height, sc.c.samples_per_pixel = w, h, 1
        tiles = rdist.shard(rank, world)
        # host_staged: bench.py's fall-back when RCCL does not come up — a data group of its own (here gloo again), the
        # tiles staged through a host buffer, the frame assembled in host memory
        grp = dist.new_group(backend="gloo") if host_staged else None
        pipe = rdist.FramePipeline(h, w, rank, world, torch.device("cpu"), group=grp, host_staged=host_staged)
        frames = []
        for i in range(n_frames):  # frame i = seed i: every frame differs, so a mixed-up buffer shows
            buf, done = pipe.begin(i)
            if i >= pipe.depth:
                frames.append(done)
            else:
                assert done is None
            sc.c.seed = i
            rgb, _, _ = oracle.render(pkg.abi, sc.ptr, tiles, n_threads=2, want_linear=False)
            buf.zero_()
            buf[: rgb.shape[0]] = torch.from_numpy(rgb)
            pipe.submit(i)
        frames += pipe.drain()
        assert len(frames) == n_frames
        if rank == 0:
            np.save(out_path, np.stack([f.numpy() for f in frames]))
        else:
            assert all(f is None for f in frames)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("host_staged", [False, True])
def test_two_rank_frame_pipeline_keeps_frames_apart(tmp_path, oracle, abi, load_scene, host_staged):
    """bench.py's N>1 loop: double-buffered tiles, asynchronous gather of frame i under frame i+1 — on the default group,
    and in the host-staged form on a data group of its own (what bench.py falls back to when RCCL does not come up)."""
    w, h, n = 24, 10, 5
    out = str(tmp_path / "frames.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), w, h, n, out, host_staged), nprocs=2, join=True)
    got = np.load(out)
    for i in range(n):
        sc = load_scene("cover", w, h, 1, seed=i)
        full, _, _ = oracle.render(abi, sc.ptr, want_linear=False)
        assert np.array_equal(got[i], full), i
