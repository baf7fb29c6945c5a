#!/bin/bash
# The lit kernels' census (round 6): build/ab/librt_hip_prof_lit.so = -DRT_PROFILE -DRT_PROF_LIT -DRT_TEST_PROBES (rt_kernel.hip):
# shader-clock cycles per wave iteration of lane_shade's parts and how often some lane of a wave takes each light continuation.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DRT_WAVES_PER_EU=4 -DRT_TEST_PROBES -DRT_PROFILE -DRT_PROF_LIT -shared \
#         rust-raytracer_amd/csrc/hip/rt_hip_api.hip -o build/ab/librt_hip_prof_lit.so
cd "${GRAFT_REPO_ROOT:-.}"
LIB=build/ab/librt_hip_prof_lit.so
python - <<'PY'
import json
j = json.load(open("scenes/cfg2_cover_1200x800_spp128.json")); j["samples_per_pixel"] = 32
json.dump(j, open("build/ab/cover_spp32.json", "w"))
j["objects"].append({"center": {"x": 0.0, "y": 30.0, "z": 10.0}, "radius": 8.0, "material": {"Light": {}}})
json.dump(j, open("build/ab/lit_cover_spp32.json", "w"))
PY
{
echo "# $LIB: cycles per wave iteration — outside lane_shade | shade: decide | ACT_SAMPLE | ACT_RETURN | FINISH/CONTINUE tail; share of iterations with SOME lane in ACT_SAMPLE / ACT_RETURN; lanes per such iteration"
for S in "--scene build/ab/cover_spp32.json --opt force_lit=1" "--scene build/ab/lit_cover_spp32.json" "--scene scenes/cfg1_test_800x600_spp16.json"; do
  timeout 100 python tools/diag.py --lib $LIB $S --reps 4 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
pc = d['prof_cycles_per_wave_iter']; wi = d['wave_iters']
names = ['refill', 'large', 'lane_shade', 'walk', 'accumulate', 'item']
c = [pc[n] for n in names]
lanes = int(round(c[5] * wi[0]))
ls, lr = lanes & 0xFFFFFFFF, lanes >> 32
tot = pc['total']
print('$S'.split('/')[-1].split()[0], '| kernel_ms', d['kernel_ms'], '| cycles/iter total %.0f: outside %.0f, decide %.0f, ACT_SAMPLE %.0f (%.1f %%), ACT_RETURN %.0f (%.1f %%), tail %.0f' % (tot, c[0], c[1], c[2], 100 * c[2] / tot, c[3], 100 * c[3] / tot, c[4]),
      '| iterations with some lane in ACT_SAMPLE %.3f, in ACT_RETURN %.3f' % (wi[1] / wi[0], wi[2] / wi[0]),
      '| lanes per such iteration %.2f / %.2f' % (ls / max(1, wi[1]), lr / max(1, wi[2])),
      '| cycles per iteration that takes it: SAMPLE %.0f, RETURN %.0f' % (c[2] * wi[0] / max(1, wi[1]), c[3] * wi[0] / max(1, wi[2])))
"
done
} | tee gpurun_out/lit_census.log
