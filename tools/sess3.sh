#!/bin/bash
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/summary.txt
T0=$(date +%s)
stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))s" | tee -a $OUT/summary.txt; }
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; stamp pytest $?
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3
AB_EXTRA="round3 cull_landing" timeout 900 bash tools/ab_scenes.sh > /dev/null 2>&1; stamp ab_scenes $?
cat $OUT/ab_scenes.log
{ echo "== lit cover spp32"; timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/lit_cover_spp32.json --only warmup_default default round3 cull_landing; } 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_lit_cover.log
{
for L in prof cull_landing_prof; do
  echo -n "$L headline: "; timeout 60 python tools/diag.py --lib build/ab/librt_hip_$L.so --reps 4 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d[k] for k in ('kernel_ms', 'exact_per_segment', 'steps_per_segment', 'wave_step_iters_per_wave_iter', 'wave_test_iters_per_wave_iter', 'lane_util_steps', 'lane_util_tests', 'prof_cycles_per_wave_iter') if k in d}))"
done
} 2>&1 | tee $OUT/cull_landing_rounds.log
echo done | tee -a $OUT/summary.txt
