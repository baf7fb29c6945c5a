#!/bin/bash
# round 5, GPU session 3: the whole GPU suite on ABI v5 (pools, transport fallback, staging, probe library), lit A/B, group overheads, bench
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/summary.txt
T0=$(date +%s)
stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))s" | tee -a $OUT/summary.txt; }
timeout 240 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; stamp smoke $?
timeout 1200 python -m pytest tests -m gpu -q -s --maxfail=8 --durations=8 > $OUT/pytest_gpu.log 2>&1; stamp pytest $?
grep -E "passed|failed|FAILED|^E  " $OUT/pytest_gpu.log | head -40
grep -E "group transport case|submit\(out = pageable|lit cover pools" $OUT/pytest_gpu.log | cut -c1-400
timeout 500 bash tools/ab_lit.sh > /dev/null 2>&1; stamp ab_lit $?
cat $OUT/ab_lit.log | grep -v "^$" | cut -c1-300
timeout 300 python tools/group_overhead.py --reps 30 > $OUT/group_overhead.json 2>$OUT/group_overhead.err; stamp group_overhead $?
cut -c1-700 $OUT/group_overhead.json
RT_GPUS_EMULATE=1 timeout 300 python bench.py --gpus 8 --steps 5 --warmup 2 > $OUT/bench_group8.log 2>$OUT/bench_group8.err; stamp bench_group8 $?
tail -1 $OUT/bench_group8.log | cut -c1-2500
RT_GPUS_EMULATE=1 RT_GROUP_PIN=0 timeout 300 python bench.py --gpus 8 --steps 5 --warmup 2 > $OUT/bench_group8_nopin.log 2>/dev/null; stamp bench_group8_nopin $?
tail -1 $OUT/bench_group8_nopin.log | cut -c1-600
timeout 400 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1; stamp bench $?
tail -1 $OUT/bench.log | cut -c1-3000
echo done | tee -a $OUT/summary.txt
