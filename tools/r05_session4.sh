#!/bin/bash
# round 5, GPU session 4: does the code LAYOUT of the module move the headline kernel?  (same kernel source: with / without the probe kernels in the
# module, loops aligned to 32 / 64 bytes) against the round-4 build, twice, arms in two orders
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
{
echo "== headline, order A"; timeout 300 python tools/ab_bench.py run --rounds 7 --only prev default probes align32 align64
echo "== headline, order A again"; timeout 300 python tools/ab_bench.py run --rounds 7 --only prev default probes align32 align64
echo "== cover + 1 light at spp 32"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/lit_cover_spp32.json --only prev default probes align32 align64
echo "== cfg1"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 15 --scene scenes/cfg1_test_800x600_spp16.json --only prev default probes align32 align64
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_layout.log
timeout 300 python -m pytest tests -m gpu -q -s -k "statistics or orientation" 2>&1 | tail -8
