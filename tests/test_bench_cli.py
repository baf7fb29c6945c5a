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
