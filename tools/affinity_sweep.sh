cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp; REPO=$PWD
{
for R in 7 8 9 10 11; do
  export RT_AFF_RUN_LOG2=$R
  echo -n "run_log2=$R whole "; timeout 60 python tools/diag.py --reps 10 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
  echo -n "run_log2=$R cfg3@spp32 "; timeout 60 python tools/diag.py --scene scenes/cfg3_cover_4k_textured.json --spp 32 --reps 4 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
  D=$REPO/gpurun_out/pmc_affr$R; rm -rf $D
  ( cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_w -o d -- python $REPO/tools/diag.py --reps 6 ) > /dev/null 2>&1
  python tools/pmc_summary.py $D | python -c "import sys,json; d=json.load(sys.stdin); print('run_log2=$R', {k:round(v,1) for k,v in d['mean_per_launch'].items()})"
done
unset RT_AFF_RUN_LOG2
echo -n "affinity off whole "; timeout 60 python tools/diag.py --reps 10 --opt tile_affinity=0 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
} 2>&1 | tee gpurun_out/r02_affinity_sweep.log
