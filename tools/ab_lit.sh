#!/bin/bash
# interleaved A/B of two megakernel builds (tools/ab_bench.py arms, default: prev vs default) on the LIT frames: cover + one light at
# spp 32, the reference's test_scene (cfg1), and the headline frame as the unlit control.  AB_ALLOW_DIFFERENT=1: an arm that changes
# the RNG addressing renders a different (equally valid) image.
cd "${GRAFT_REPO_ROOT:-.}"
ARMS="${AB_ARMS:-prev default}"
mkdir -p build/ab gpurun_out
{
python - <<'PY'
import json
j = json.load(open("scenes/cfg2_cover_1200x800_spp128.json")); j["samples_per_pixel"] = 32
j["objects"].append({"center": {"x": 0.0, "y": 30.0, "z": 10.0}, "radius": 8.0, "material": {"Light": {}}})
json.dump(j, open("build/ab/lit_cover_spp32.json", "w"))
PY
echo "== cover + 1 light at spp 32"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/lit_cover_spp32.json --only $ARMS
echo "== cfg1"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 15 --scene scenes/cfg1_test_800x600_spp16.json --only $ARMS
echo "== headline (unlit control)"; timeout 200 python tools/ab_bench.py run --rounds 5 --only $ARMS
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_lit.log
