#!/bin/bash
# Instruction-cache and instruction-fetch counters of the megakernel (rocprofv3 --pmc, counters only): does the 29 KB unlit
# kernel / the 40 KB lit kernel fit the instruction cache two CUs share, and how long do waves wait for instructions?
# Usage on the GPU box: bash tools/pmc_icache.sh <tag>  ->  gpurun_out/<tag>_pmc_icache.json
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD; export TMPDIR=/tmp
TAG=${1:-rXX}
mkdir -p build/ab gpurun_out
python - <<'PY'
import json
j = json.load(open("scenes/cfg2_cover_1200x800_spp128.json")); j["samples_per_pixel"] = 32
json.dump(j, open("build/ab/cover_spp32.json", "w"))
j["objects"].append({"center": {"x": 0.0, "y": 30.0, "z": 10.0}, "radius": 8.0, "material": {"Light": {}}})
json.dump(j, open("build/ab/lit_cover_spp32.json", "w"))
PY
run() {  # key, diag.py arguments...
  KEY=$1; shift
  D=$REPO/gpurun_out/icache_${TAG}_$KEY; rm -rf $D; mkdir -p $D
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $D/pmc_ic -o d -- python $REPO/tools/diag.py "$@" --reps 2 ) > $D/ic.log 2>&1; echo "$KEY pass ic rc=$?"
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY --output-format csv -d $D/pmc_if -o d -- python $REPO/tools/diag.py "$@" --reps 2 ) > $D/if.log 2>&1; echo "$KEY pass if rc=$?"
  python tools/pmc_summary.py $D "$KEY: $*" > gpurun_out/${TAG}_pmc_icache_$KEY.json 2>$D/summary.err
  python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_pmc_icache_$KEY.json'))
print('$KEY', d.get('kernel_ms'), {k: round(v / 1e6, 3) for k, v in sorted(d['mean_per_launch'].items())})
PY
}
run headline
run cover_spp32 --scene $REPO/build/ab/cover_spp32.json
run cover_spp32_lit_kernel --scene $REPO/build/ab/cover_spp32.json --opt force_lit=1
run lit_cover_spp32 --scene $REPO/build/ab/lit_cover_spp32.json
run cfg1 --scene $REPO/scenes/cfg1_test_800x600_spp16.json
