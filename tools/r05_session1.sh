#!/bin/bash
# round 5, GPU session 1: the lit kernels without their scratch object — tests, A/B against the round-4 kernel, pool sweep, counters
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/summary.txt
T0=$(date +%s)
stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))s" | tee -a $OUT/summary.txt; }
timeout 240 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; stamp smoke $?
tail -3 $OUT/smoke.log
timeout 900 python -m pytest tests -m gpu -q -s --maxfail=6 --durations=8 > $OUT/pytest_gpu.log 2>&1; stamp pytest $?
tail -15 $OUT/pytest_gpu.log
timeout 500 bash tools/ab_lit.sh > /dev/null 2>&1; stamp ab_lit $?
cat $OUT/ab_lit.log | grep -v "^$" | cut -c1-300
timeout 300 python tools/pool_sweep.py > $OUT/pool_sweep.log 2>&1; stamp pool_sweep $?
cat $OUT/pool_sweep.log | cut -c1-250
timeout 120 python tools/lit_bench.py > $OUT/lit_bench.log 2>&1; stamp lit_bench $?
cat $OUT/lit_bench.log | cut -c1-250
export TMPDIR=/tmp
timeout 900 bash tools/pmc_scene.sh r05_run1 cfg1=scenes/cfg1_test_800x600_spp16.json litcover=build/ab/lit_cover_spp32.json > $OUT/pmcs.log 2>&1; stamp pmcs $?
grep -E "^(cfg1|litcover) " $OUT/pmcs.log | cut -c1-900
echo done | tee -a $OUT/summary.txt
