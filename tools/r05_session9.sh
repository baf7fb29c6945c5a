#!/bin/bash
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
for W in 1 0; do
  for S in scenes/cfg1_test_800x600_spp16.json scenes/cfg2_cover_1200x800_spp128.json; do
    for i in 1 2 3 4; do
      if [ $W = 0 ]; then export RT_NO_KERNEL_WARMUP=1; else unset RT_NO_KERNEL_WARMUP; fi
      echo -n "kernel_warmup=$W $(basename $S) "; RT_STATS=1 ./rust-raytracer_amd/raytracer $S /tmp/out.png 2>&1 >/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('kernel_ms','frame_ms','setup_ms','hip_init_ms','main_ms')}, [round(x) for x in d['group_us']])"
    done
  done
done | tee $OUT/cli_kernel_warmup.log
timeout 600 python -m pytest tests -m gpu -q -x -k "group or cli or animation or host_buffer or bench" 2>&1 | tail -5
