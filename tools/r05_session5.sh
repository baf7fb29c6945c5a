#!/bin/bash
# round 5, GPU session 5: kernel arguments back in the round-4 layout (the light overflow pointer where the dead cull table sat) — A/B, new tests, CLI stats
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
{
echo "== headline"; timeout 300 python tools/ab_bench.py run --rounds 7 --only prev default
echo "== headline again"; timeout 300 python tools/ab_bench.py run --rounds 7 --only prev default
echo "== cover + 1 light at spp 32"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/lit_cover_spp32.json --only prev default
echo "== cfg1"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 15 --scene scenes/cfg1_test_800x600_spp16.json --only prev default
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_layout2.log
timeout 600 python -m pytest tests -m gpu -q -s -k "statistics or orientation or light_pools or lit_cover or many_lights or golden or cli" 2>&1 | tail -12
for S in scenes/cfg2_cover_1200x800_spp128.json scenes/cfg1_test_800x600_spp16.json; do for i in 1 2 3; do RT_STATS=1 ./rust-raytracer_amd/raytracer $S /tmp/out.png 2>&1 >/dev/null | tail -1; done; done | tee $OUT/cli_stats.log
