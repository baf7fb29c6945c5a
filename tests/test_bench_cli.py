"""bench.py's contract on a box WITHOUT a GPU: whatever fails, stdout carries ONE JSON line with an "error" key and the exit
code is non-zero — the driver never has to parse a traceback."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_bench_without_a_gpu_prints_a_json_error_line(gpus):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode != 0
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == gpus and "GPU" in d["error"] and d["metric"].startswith("Msamples/sec")


def test_multi_gpu_line_helpers_price_ranks_from_committed_counters():
    """bench.py's N > 1 extras that need no GPU: the per-rank roofline from the committed counter files (the whole frame's lane
    cycles split over the ranks, each over its own kernel time) and the cpu_baseline pointer to the newest committed N = 1 line."""
    sys.path.insert(0, ROOT)
    import bench
    # an 8-rank frame whose shards take 1/7.45 of the whole frame's 12.73 ms: per-rank fraction ~ the N = 1 fraction x 7.45 / 8
    ks = [12.73 / 7.45] * 7 + [12.73 / 7.0]
    rf = bench._group_roofline("cfg2", 8, ks)
    assert rf["bound"] == "valu" and rf["peak"] == round(bench.PEAK_LANE_SLOTS_T, 2)
    assert len(rf["per_rank_frac"]) == 8 and 0.35 < rf["frac"] < 0.5 and rf["frac"] == min(rf["per_rank_frac"])   # frac = the slowest rank's
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-3 and rf["counters"]["source"].startswith("profiles/")
    c4 = bench._group_roofline("cfg4", 8, [470.0 / 7.6] * 8)          # BASELINE configs[3]: counters of the 4K textured spp-512 frame
    assert c4["frac"] is not None and 0.3 < c4["frac"] < 0.5 and "pmc_cfg4" in c4["counters"]["source"]
    assert bench._group_roofline("cfg2", 4, [])["frac"] is None           # nothing measured: nothing claimed
    cb = bench._n1_cpu_baseline_pointer()
    assert cb is not None and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["source"].startswith("profiles/") and "bench.json" in cb["source"]
