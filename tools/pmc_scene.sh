#!/bin/bash
# PMC counters of the megakernel on arbitrary scenes (one rocprofv3 pass per counter set per scene; counters only).
# Usage on the GPU box: bash tools/pmc_scene.sh <tag> <scene.json> [<scene.json> ...]  -> gpurun_out/pmcs_<tag>_<scene>.json
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD; export TMPDIR=/tmp
TAG=$1; shift
for S in "$@"; do
  B=$(basename $S .json); D=$REPO/gpurun_out/pmcs_${TAG}_$B; rm -rf $D; mkdir -p $D
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $D/pmc_a -o d -- python $REPO/tools/diag.py --scene $REPO/$S --reps 3 ) > $D/a.log 2>&1
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $D/pmc_b -o d -- python $REPO/tools/diag.py --scene $REPO/$S --reps 3 ) > $D/b.log 2>&1
  ( cd /tmp && timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC --output-format csv -d $D/pmc_c -o d -- python $REPO/tools/diag.py --scene $REPO/$S --reps 3 ) > $D/c.log 2>&1
  python tools/pmc_summary.py $D > gpurun_out/pmcs_${TAG}_$B.json 2>/dev/null
  tail -2 $D/c.log | cut -c1-300
  python -c "
import json;d=json.load(open('gpurun_out/pmcs_${TAG}_$B.json'))['mean_per_launch'];print('$B',{k:round(v/1e6,2) for k,v in sorted(d.items())})"
done
