cd "${GRAFT_REPO_ROOT:-.}"
for O in "" "--opt force_lit=1"; do for i in 1 2; do echo -n "cover spp32 unlit scene [$O] "; timeout 60 python tools/diag.py --scene build/ab/cover_spp32.json --reps 10 $O 2>/dev/null | tail -1 | cut -c1-200; done; done
for O in "" "--opt force_lit=1"; do echo -n "headline [$O] "; timeout 60 python tools/diag.py --reps 6 $O 2>/dev/null | tail -1 | cut -c1-200; done
echo -n "lit cover "; timeout 60 python tools/diag.py --scene build/ab/lit_cover_spp32.json --reps 10 2>/dev/null | tail -1 | cut -c1-200
