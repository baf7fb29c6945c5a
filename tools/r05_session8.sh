#!/bin/bash
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
for W in 1 0; do
  for S in scenes/cfg1_test_800x600_spp16.json scenes/cfg2_cover_1200x800_spp128.json; do
    for i in 1 2 3; do
      if [ $W = 0 ]; then export RT_NO_COPY_WARMUP=1; else unset RT_NO_COPY_WARMUP; fi
      echo -n "copy_warmup=$W $(basename $S) "; RT_GROUP_TRACE=1 RT_STATS=1 ./rust-raytracer_amd/raytracer $S /tmp/out.png 2>&1 >/dev/null | tr '\n' ' ' | python -c "
import sys,json,re
t=sys.stdin.read(); tr=re.search(r'\[rt group\][^{]*', t); d=json.loads(t[t.index('{'):t.rindex('}')+1])
print((tr.group(0).strip() if tr else ''), {k:d[k] for k in ('kernel_ms','frame_ms','setup_ms','main_ms')}, [round(x) for x in d['group_us']])"
    done
  done
done | tee $OUT/cli_copy_warmup.log
