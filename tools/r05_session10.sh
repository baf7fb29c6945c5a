#!/bin/bash
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests -m gpu -q -x -s -k "two_ranks or group or cli or animation or host_buffer or bench" > gpurun_out/pytest_sub.log 2>&1; echo "rc=$?"
grep -E "passed|failed|FAILED|^E  |two ranks on one GPU" gpurun_out/pytest_sub.log | cut -c1-400 | head -20
