cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp; REPO=$PWD
{
timeout 200 python tools/ab_bench.py run --rounds 9 --only prevsrc aff_off aff_on 2>&1 | grep -v amdgpu.ids
for A in 0 1; do
  echo -n "tile_affinity=$A shard 3/8 "; timeout 60 python tools/diag.py --shard 3,8,2 --reps 8 --opt tile_affinity=$A 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
  echo -n "tile_affinity=$A cfg3@spp32 "; timeout 60 python tools/diag.py --scene scenes/cfg3_cover_4k_textured.json --spp 32 --reps 4 --opt tile_affinity=$A 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
  D=$REPO/gpurun_out/pmc_aff$A; rm -rf $D
  ( cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_w -o d -- python $REPO/tools/diag.py --reps 6 --opt tile_affinity=$A ) > /dev/null 2>&1
  ( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_f -o d -- python $REPO/tools/diag.py --reps 6 --opt tile_affinity=$A ) > /dev/null 2>&1
  python tools/pmc_summary.py $D | python -c "import sys,json; d=json.load(sys.stdin); print('tile_affinity=$A', {k:round(v,1) for k,v in d['mean_per_launch'].items()}, d.get('hbm_bytes_per_launch'))"
done
} 2>&1 | tee gpurun_out/r02_affinity.log
