#!/bin/bash
# tile size x chunk size on a 4K frame (cfg3 geometry at spp 128): kernel ms, best of 4
cd "${GRAFT_REPO_ROOT:-.}"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"; }
{
echo -n "auto: "; timeout 100 python tools/diag.py --scene scenes/cfg3_cover_4k_textured.json --spp 128 --reps 4 2>/dev/null | tail -1 | ms
for TL in 2 3; do for CS in 4 8 16 32 64 128; do
  echo -n "tile_log2=$TL chunk_spp=$CS: "; timeout 100 python tools/diag.py --scene scenes/cfg3_cover_4k_textured.json --spp 128 --reps 4 --opt tile_log2=$TL chunk_spp=$CS 2>/dev/null | tail -1 | ms
done; done
for AFF in 0 1; do echo -n "tile_affinity=$AFF: "; timeout 100 python tools/diag.py --scene scenes/cfg3_cover_4k_textured.json --spp 128 --reps 4 --opt tile_affinity=$AFF 2>/dev/null | tail -1 | ms; done
for SH in 0 2 3; do echo -n "tile_shape=$SH: "; timeout 100 python tools/diag.py --scene scenes/cfg3_cover_4k_textured.json --spp 128 --reps 4 --opt tile_shape=$SH 2>/dev/null | tail -1 | ms; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tile_sweep_4k.log
