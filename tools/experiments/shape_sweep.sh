cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp; REPO=$PWD
{
for SH in 0 2 3 1; do
  echo -n "tile_shape=$SH whole "; timeout 60 python tools/diag.py --reps 8 --opt tile_shape=$SH 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
  echo -n "tile_shape=$SH shard 3/8 "; timeout 60 python tools/diag.py --shard 3,8,2 --reps 8 --opt tile_shape=$SH 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
  D=$REPO/gpurun_out/pmc_shape$SH; rm -rf $D
  ( cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_w -o d -- python $REPO/tools/diag.py --reps 2 --opt tile_shape=$SH ) > /dev/null 2>&1
  ( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_f -o d -- python $REPO/tools/diag.py --reps 2 --opt tile_shape=$SH ) > /dev/null 2>&1
  python tools/pmc_summary.py $D | python -c "import sys,json; d=json.load(sys.stdin); print('tile_shape=$SH', {k:round(v,1) for k,v in d['mean_per_launch'].items()}, d.get('hbm_bytes_per_launch'))"
done
echo "# cfg3 at spp 32: section profile (prof build), then the untextured cover at the same frame size"
timeout 100 python tools/diag.py --lib build/ab/librt_hip_prof.so --scene scenes/cfg3_cover_4k_textured.json --spp 32 --reps 3 2>/dev/null | tail -1
timeout 100 python tools/diag.py --lib build/ab/librt_hip_prof.so --width 3840 --height 2160 --spp 32 --reps 3 2>/dev/null | tail -1
} 2>&1 | tee gpurun_out/r02_shape_sweep.log
