#!/bin/bash
# PMC counters of the megakernel on arbitrary scenes (one rocprofv3 pass per counter set per scene; counters only —
# never combined with trace domains).  HBM traffic as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (TCC slots), FETCH_SIZE x 2 on gfx950; L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum).
# Usage on the GPU box: bash tools/pmc_scene.sh <tag> <key>=<scene.json | procedural:HALF:SPP> ...
#   -> gpurun_out/<tag>_pmc_<key>.json  (self-describing: scene, kernel_ms of the passes, git_head)
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$PWD; export TMPDIR=/tmp
TAG=$1; shift
for KV in "$@"; do
  KEY=${KV%%=*}; S=${KV#*=}
  if [[ $S == procedural:* ]]; then IFS=: read -r _ HALF SPP <<< "$S"; ARGS="--procedural $HALF --spp $SPP"; else ARGS="--scene $REPO/$S"; fi
  D=$REPO/gpurun_out/pmcs_${TAG}_$KEY; rm -rf $D; mkdir -p $D
  pass() {  # name counters...
    N=$1; shift
    ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $D/pmc_$N -o d -- python $REPO/tools/diag.py $ARGS --reps 2 ) > $D/$N.log 2>&1
    echo "$KEY pass $N rc=$?"
  }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass l2 TCC_HIT_sum TCC_MISS_sum
  pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  pass b SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
  pass c SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT
  python tools/pmc_summary.py $D "$KEY: $S" > gpurun_out/${TAG}_pmc_$KEY.json 2>$D/summary.err
  python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_pmc_$KEY.json'))
m = d['mean_per_launch']
print('$KEY', {k: d.get(k) for k in ('kernel_ms', 'git_head', 'hbm_bytes_per_launch', 'l2_hit_rate', 'lane_utilisation', 'lane_slot_frac_at_that_time')})
print('$KEY', {k: round(v / 1e6, 3) for k, v in sorted(m.items())})
PY
done
