#!/bin/bash
# interleaved A/B (tools/ab_bench.py arms) on the headline frame, the textured 4K frame at spp 64 and the lit test scene
cd "${GRAFT_REPO_ROOT:-.}"
{
echo "== headline"; timeout 200 python tools/ab_bench.py run --rounds 7 --only warmup_default default ${AB_EXTRA}
python - <<'PY'
import json
j = json.load(open("scenes/cfg3_cover_4k_textured.json")); j["samples_per_pixel"] = 64
json.dump(j, open("build/ab/cfg3_spp64.json", "w"))
j = json.load(open("scenes/cfg2_cover_1200x800_spp128.json")); j["samples_per_pixel"] = 32
json.dump(j, open("build/ab/cover_spp32.json", "w"))
PY
echo "== cfg3 at spp 64"; timeout 300 python tools/ab_bench.py run --rounds 5 --scene build/ab/cfg3_spp64.json --only warmup_default default ${AB_EXTRA}
echo "== cover at spp 32"; timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/cover_spp32.json --only warmup_default default ${AB_EXTRA}
echo "== cfg1"; timeout 200 python tools/ab_bench.py run --rounds 9 --scene scenes/cfg1_test_800x600_spp16.json --only warmup_default default ${AB_EXTRA}
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_scenes.log
