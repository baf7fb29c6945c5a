#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench line, rocprofv3 kernel stats (+ optional A/B).
# Usage (from the build container):  gpurun --timeout 900 -- 'bash tools/gpu_check.sh [ab] [pmc]'
# Every step runs under `timeout` and nothing reads stdin: a stuck step must not eat GPU budget.
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/summary.txt
T0=$(date +%s)
stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))s" | tee -a $OUT/summary.txt; }
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 240 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; stamp smoke $?
timeout 600 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; stamp pytest $?
tail -4 $OUT/pytest_gpu.log
if [[ " $* " == *" ab "* ]]; then
  timeout 300 python tools/ab_bench.py run --rounds 5 > $OUT/ab.log 2>&1; stamp ab $?
  cat $OUT/ab.log | tail -20
fi
timeout 300 python bench.py --steps 5 --warmup 1 > $OUT/bench.log 2>&1; stamp bench $?
tail -1 $OUT/bench.log
export TMPDIR=/tmp
REPO=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/prof" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1
stamp rocprof $?
STATS=$(find $OUT/prof -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$STATS" ]; then head -6 "$STATS"; fi
if [[ " $* " == *" pmc "* ]]; then
  # HBM traffic: separate PMC passes, counters only (no trace domains), per the microarch guide
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$REPO/$OUT/pmc_fetch" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ) > $OUT/pmc_fetch.log 2>&1; stamp pmc_fetch $?
  ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$REPO/$OUT/pmc_write" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ) > $OUT/pmc_write.log 2>&1; stamp pmc_write $?
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$REPO/$OUT/pmc_sq" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ) > $OUT/pmc_sq.log 2>&1; stamp pmc_sq $?
fi
echo done | tee -a $OUT/summary.txt
