#!/bin/bash
# round 5, GPU session 6: overlapped vs sequential pool takes, flush_tile variants, one-shot CLI frames after the warm-up at scene creation
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
{
echo "== headline"; timeout 300 python tools/ab_bench.py run --rounds 7 --only prev default opaque
echo "== headline again"; timeout 300 python tools/ab_bench.py run --rounds 7 --only prev default opaque
echo "== cover + 1 light at spp 32"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/lit_cover_spp32.json --only prev default seq opaque
echo "== cfg1"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 15 --scene scenes/cfg1_test_800x600_spp16.json --only prev default seq opaque
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_takes.log
for S in scenes/cfg2_cover_1200x800_spp128.json scenes/cfg1_test_800x600_spp16.json; do for i in 1 2 3; do RT_STATS=1 ./rust-raytracer_amd/raytracer $S /tmp/out.png 2>&1 >/dev/null | tail -1; done; done | tee $OUT/cli_stats.log
timeout 600 python -m pytest tests -m gpu -q -x -k "light or lit or lights or golden or cfg1 or cli or smoke or stress" 2>&1 | tail -5
