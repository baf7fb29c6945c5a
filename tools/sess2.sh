#!/bin/bash
# round-4 session 2: full GPU tests after the scene-create fix; A/B default vs -disable-machine-licm vs round 3; lit bench; first-frame timelines
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/summary.txt
T0=$(date +%s)
stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))s" | tee -a $OUT/summary.txt; }
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=5 > $OUT/pytest_gpu.log 2>&1; stamp pytest $?
tail -12 $OUT/pytest_gpu.log
AB_EXTRA="round3 nolicm" timeout 900 bash tools/ab_scenes.sh > /dev/null 2>&1; stamp ab_scenes $?
cat $OUT/ab_scenes.log
{
echo "== lit cover spp32 A/B"
python - <<'PY'
import json, sys
sys.path.insert(0, "tools")
from lit_bench import cover
open("build/ab/lit_cover_spp32.json", "w").write(cover(32, light=True))
PY
timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/lit_cover_spp32.json --only warmup_default default round3 nolicm
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_lit_cover.log
timeout 200 python tools/lit_bench.py > $OUT/lit_bench.log 2>&1; stamp lit_bench $?
grep -v amdgpu.ids $OUT/lit_bench.log
timeout 200 python tools/lit_bench.py --lib build/ab/librt_hip_nolicm.so > $OUT/lit_bench_nolicm.log 2>&1; stamp lit_bench_nolicm $?
grep -v amdgpu.ids $OUT/lit_bench_nolicm.log
{
echo "# first frame of a scene (RT_PROFILE build, --reps 1: a fresh scene each): bottom row first / projection seed / probe seed; then the steady state (3 frames)"
for O in "tile_order=1" "tile_order=3 order_seed=1" "tile_order=3 order_seed=2"; do
  for i in 1 2; do echo -n "$O: "; timeout 60 python tools/diag.py --lib build/ab/librt_hip_prof.so --reps 1 --opt $O 2>/dev/null | tail -1; done
done
echo -n "steady: "; timeout 60 python tools/diag.py --lib build/ab/librt_hip_prof.so --reps 4 2>/dev/null | tail -1
echo "# the same on an 1/8 shard (rank 3)"
for O in "tile_order=1" "tile_order=3 order_seed=1" "tile_order=3 order_seed=2"; do
  echo -n "$O: "; timeout 60 python tools/diag.py --lib build/ab/librt_hip_prof.so --shard 3,8,2 --reps 1 --opt $O 2>/dev/null | tail -1
done
echo -n "steady: "; timeout 60 python tools/diag.py --lib build/ab/librt_hip_prof.so --shard 3,8,2 --reps 4 2>/dev/null | tail -1
} > $OUT/first_frame_timelines.log 2>&1; stamp first_frame $?
python - <<'PY'
import json
for line in open("gpurun_out/first_frame_timelines.log"):
    if "{" not in line:
        print(line.strip()); continue
    name, js = line.split(": {", 1)
    d = json.loads("{" + js)
    t = d.get("timeline_us", {})
    print(name, "kernel_ms", d["kernel_ms"], "end_pct", t.get("end_pct"), "idle_after_end", t.get("idle_frac_after_end"), "tail_iters", t.get("tail", {}).get("tail_iters_pct"), "queue_empty_seen", t.get("tail", {}).get("queue_empty_seen_pct"))
PY
echo done | tee -a $OUT/summary.txt
