cd "${GRAFT_REPO_ROOT:-.}"
{
echo -n "auto "; timeout 60 python tools/diag.py --scene scenes/cfg1_test_800x600_spp16.json --reps 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
for TL in 1 2 3; do for CS in 1 2 4 8 16; do
  echo -n "tile_log2=$TL chunk_spp=$CS "; timeout 60 python tools/diag.py --scene scenes/cfg1_test_800x600_spp16.json --reps 8 --opt tile_log2=$TL chunk_spp=$CS 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
done; done
echo -n "unlit cover 800x600 spp16 auto "; timeout 60 python tools/diag.py --width 800 --height 600 --spp 16 --reps 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
} 2>&1 | tee gpurun_out/r02_cfg1_sweep.log
