#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench line, rocprofv3 kernel stats.
# Usage (from the build container):  gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 1 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/bench.log
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1
echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*stats*" | head; cat $(find $OUT/prof -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -8
