cd "${GRAFT_REPO_ROOT:-.}"
P=build/ab/librt_hip_prof.so
for C in "build/ab/cover_spp32.json|" "build/ab/cover_spp32.json|force_lit=1 light_takers=16" "build/ab/lit_cover_spp32.json|light_takers=16" "build/ab/lit_cover_spp32.json|light_takers=4"; do
  S=${C%%|*}; O=${C##*|}
  echo -n "$S [$O]: "
  timeout 60 python tools/diag.py --lib $P --scene $S --reps 4 ${O:+--opt $O} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d[k] for k in ('kernel_ms', 'segments_per_sample', 'lane_util_segments', 'wave_iters', 'wave_step_iters_per_wave_iter', 'wave_test_iters_per_wave_iter', 'prof_cycles_per_wave_iter') if k in d}))"
done
