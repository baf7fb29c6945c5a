#!/bin/bash
# round-4 session 1: tests, group overhead, lit baseline, A/B against the round-3 kernel, bench
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/summary.txt
T0=$(date +%s)
stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))s" | tee -a $OUT/summary.txt; }
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 240 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; stamp smoke $?
tail -3 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; stamp pytest $?
tail -15 $OUT/pytest_gpu.log
timeout 300 python tools/group_overhead.py --reps 40 > $OUT/group_overhead.json 2>$OUT/group_overhead.err; stamp group_overhead $?
cut -c1-900 $OUT/group_overhead.json
timeout 200 python tools/lit_bench.py > $OUT/lit_bench.log 2>&1; stamp lit_bench $?
grep -v amdgpu.ids $OUT/lit_bench.log
AB_EXTRA="round3" timeout 900 bash tools/ab_scenes.sh > /dev/null 2>&1; stamp ab_scenes $?
cat $OUT/ab_scenes.log
timeout 400 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>$OUT/bench.err; stamp bench $?
tail -1 $OUT/bench.log
RT_GPUS_EMULATE=1 timeout 300 python bench.py --gpus 8 --steps 10 --warmup 2 > $OUT/bench_group8.log 2>$OUT/bench_group8.err; stamp bench_group8 $?
tail -1 $OUT/bench_group8.log | cut -c1-2500
echo done | tee -a $OUT/summary.txt
