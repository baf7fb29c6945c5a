#!/bin/bash
# Tile shape vs kernel time and WRITE_SIZE, whole headline frame and one 1/8 shard (product library, no dev knobs).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp; REPO=$PWD
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"; }
wr() {
  D=$REPO/gpurun_out/pmc_shape; rm -rf $D
  ( cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_w -o d -- python $REPO/tools/diag.py --reps 6 "$@" ) > /dev/null 2>&1
  python tools/pmc_summary.py $D | python -c "import sys,json; d=json.load(sys.stdin); print('WRITE_SIZE KiB', round(d['mean_per_launch'].get('WRITE_SIZE', -1), 1))"
}
{
for SHAPE in 0 2 3 1; do
  echo -n "shard 3,8,2 tile_shape=$SHAPE: "; timeout 60 python tools/diag.py --shard 3,8,2 --reps 10 --opt tile_shape=$SHAPE 2>/dev/null | tail -1 | ms
  echo -n "shard 3,8,2 tile_shape=$SHAPE: "; wr --shard 3,8,2 --opt tile_shape=$SHAPE
  echo -n "whole tile_shape=$SHAPE: "; timeout 60 python tools/diag.py --reps 10 --opt tile_shape=$SHAPE 2>/dev/null | tail -1 | ms
  echo -n "whole tile_shape=$SHAPE: "; wr --opt tile_shape=$SHAPE
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/shape_traffic_sweep.log
