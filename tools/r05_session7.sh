#!/bin/bash
# round 5, GPU session 7: one-shot CLI frames against the warm-up spin; final-form A/B; full GPU suite; bench line
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
for SPIN in 0 500 2000 5000 20000; do
  for S in scenes/cfg2_cover_1200x800_spp128.json scenes/cfg1_test_800x600_spp16.json; do
    for i in 1 2 3; do echo -n "spin_us=$SPIN $(basename $S) "; RT_WARM_SPIN_US=$SPIN RT_STATS=1 ./rust-raytracer_amd/raytracer $S /tmp/out.png 2>&1 >/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('kernel_ms','frame_ms','setup_ms','hip_init_ms','main_ms','group_us')})"; done
  done
done | tee $OUT/cli_spin.log
{
echo "== headline"; timeout 300 python tools/ab_bench.py run --rounds 7 --only prev default
echo "== cover + 1 light at spp 32"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 9 --scene build/ab/lit_cover_spp32.json --only prev default
echo "== cfg1"; AB_ALLOW_DIFFERENT=1 timeout 200 python tools/ab_bench.py run --rounds 15 --scene scenes/cfg1_test_800x600_spp16.json --only prev default
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_final.log
timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=8 --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |mixed radii" $OUT/pytest_gpu.log | cut -c1-400 | head -30
timeout 500 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>$OUT/bench.err; echo "bench rc=$?"
tail -1 $OUT/bench.log | cut -c1-6000
