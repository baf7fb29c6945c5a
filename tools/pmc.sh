#!/bin/bash
# PMC passes over the headline frame (counters only, separate passes; MI355X_MICROARCH.md §rocprofv3).
# usage: bash tools/pmc.sh <outdir> [diag.py args]
set -u
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
run_pass() {
  name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc $* --output-format csv -d "$REPO/$OUT/$name" -o p -- python "$REPO/tools/diag.py" --reps 2 $EXTRA ) > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
EXTRA="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run_pass sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_megakernel" not in r.get("Kernel_Name", ""):
            continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
# each dispatch contributes one row per counter (possibly per dimension): report mean per launch
launches = max(1, min(v[1] for v in agg.values())) if agg else 1
res = {k: v[0] / max(1, v[1]) * (v[1] / launches) for k, v in agg.items()}
print(json.dumps({"launches": launches, "mean_per_launch": res}, indent=1))
json.dump({"launches": launches, "mean_per_launch": res}, open(out + "/pmc_summary.json", "w"), indent=1)
PY
