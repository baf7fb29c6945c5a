#!/bin/bash
# wall time of one-shot CLI runs (process start -> exit), fast exit on / off
cd "${GRAFT_REPO_ROOT:-.}"
for FE in 1 0; do for S in scenes/cfg2_cover_1200x800_spp128.json scenes/cfg1_test_800x600_spp16.json; do for i in 1 2 3 4 5; do
  T0=$(date +%s%N); RT_FAST_EXIT=$FE RT_STATS=1 ./rust-raytracer_amd/raytracer $S /tmp/out.png 2>/tmp/err.txt >/dev/null; T1=$(date +%s%N)
  echo "fast_exit=$FE $(basename $S) wall_ms=$(( (T1-T0)/1000000 )) $(tail -1 /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('main_ms','load_ms','hip_init_ms','hip_wait_ms','setup_ms','frame_ms','png_ms')})")"
done; done; done
timeout 300 python -m pytest tests -m gpu -q -x -k "cli or animation" 2>&1 | tail -3
