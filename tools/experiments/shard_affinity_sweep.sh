#!/bin/bash
# Per-XCD tile queues on the SHARDS of a multi-GPU frame: kernel time and HBM write traffic of one rank's shard (2-scanline
# interleave) for run lengths of 2^R pixels (A/B build with -DRT_DEV_KNOBS: RT_AFF_RUN_LOG2), affinity forced on
# (tile_affinity=2) against affinity off.  Usage on the GPU box: bash tools/shard_affinity_sweep.sh
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp; REPO=$PWD
LIB=build/ab/librt_hip_default.so
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"; }
wr() {  # WRITE_SIZE of the launches of: diag.py <args>
  D=$REPO/gpurun_out/pmc_shaff; rm -rf $D
  ( cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_w -o d -- python $REPO/tools/diag.py --lib $REPO/$LIB --reps 6 "$@" ) > /dev/null 2>&1
  python tools/pmc_summary.py $D | python -c "import sys,json; d=json.load(sys.stdin); print('WRITE_SIZE KiB', round(d['mean_per_launch'].get('WRITE_SIZE', -1), 1))"
}
{
for SH in 3,8,2 0,4,2; do
  echo -n "shard $SH affinity off: "; timeout 60 python tools/diag.py --lib $LIB --shard $SH --reps 10 --opt tile_affinity=0 2>/dev/null | tail -1 | ms
  echo -n "shard $SH affinity off: "; wr --shard $SH --opt tile_affinity=0
  for R in 5 6 7 8 9; do
    export RT_AFF_RUN_LOG2=$R
    echo -n "shard $SH run 2^$R px, affinity forced: "; timeout 60 python tools/diag.py --lib $LIB --shard $SH --reps 10 --opt tile_affinity=2 2>/dev/null | tail -1 | ms
    echo -n "shard $SH run 2^$R px, affinity forced: "; wr --shard $SH --opt tile_affinity=2
    unset RT_AFF_RUN_LOG2
  done
done
for R in 6 7; do
  export RT_AFF_RUN_LOG2=$R
  echo -n "whole frame run 2^$R px: "; timeout 60 python tools/diag.py --lib $LIB --reps 10 2>/dev/null | tail -1 | ms
  echo -n "whole frame run 2^$R px: "; wr
  unset RT_AFF_RUN_LOG2
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/shard_affinity_sweep.log
