"""The workgroup tile stash of rt_kernel.hip::acquire() restated with std::atomic and stressed on host
threads (tests/hostsim/stash_model.cpp): every queue position is opened exactly once whatever the interleaving, for one
queue and for eight per-XCD queues, with batches larger than a queue, frames smaller than a batch and empty frames.  The
HIP code itself is covered on the GPU (test_work_distribution_stress: identical frames for every batch size)."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def stash_model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("stash") / "stash_model")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(HERE, "hostsim", "stash_model.cpp"), "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("n_tiles,groups,waves,batch,share,queues", [
    (1000, 4, 8, 4, 16, 1),      # the product's settings: batches of 4, taper remaining / (groups x 16)
    (1000, 4, 8, 64, 1, 8),      # batches larger than a queue's share: clipped last batches, queues seen dry, stealing
    (7, 8, 4, 4, 16, 8),         # fewer tiles than queues
    (5000, 16, 4, 3, 4, 8),
    (0, 2, 2, 4, 16, 1),         # empty frame
    (20000, 8, 8, 16, 2, 1),
    (1, 3, 5, 1, 16, 1),
])
def test_every_tile_is_opened_exactly_once(stash_model, n_tiles, groups, waves, batch, share, queues):
    for seed in range(3):
        r = subprocess.run([stash_model, str(n_tiles), str(groups), str(waves), str(batch), str(share), str(queues), str(seed)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        out = json.loads(r.stdout)
        assert out["bad"] == 0 and out["n_tiles"] == n_tiles
