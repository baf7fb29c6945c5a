cd "${GRAFT_REPO_ROOT:-.}"
P=build/ab/librt_hip_prof.so
for o in "" "--opt chunk_spp=2" "--opt chunk_spp=1"; do echo "cfg1 $o"; timeout 60 python tools/diag.py --lib $P --scene scenes/cfg1_test_800x600_spp16.json --reps 7 $o 2>/dev/null | tail -1; done
for o in "" "--opt chunk_spp=8" "--opt chunk_spp=4"; do echo "shard $o"; timeout 60 python tools/diag.py --lib $P --shard 3,8,2 --reps 7 $o 2>/dev/null | tail -1; done
for o in "" "--opt chunk_spp=8"; do echo "headline $o"; timeout 60 python tools/diag.py --lib $P --reps 5 $o 2>/dev/null | tail -1; done
