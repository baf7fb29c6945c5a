#!/bin/bash
# kernel time of one rank's shard of the headline frame at a time (2-scanline interleave), G = 1, 2, 4, 8; best of 10
cd "${GRAFT_REPO_ROOT:-.}"
for SH in "" 0,2,2 1,2,2 0,4,2 3,4,2 0,8,2 3,8,2 7,8,2; do
  echo -n "shard [$SH] ${1:-} "; timeout 60 python tools/diag.py ${SH:+--shard $SH} --reps 10 ${1:+--opt $1} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['msamples_per_s'])"
done
