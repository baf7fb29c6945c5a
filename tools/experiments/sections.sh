#!/bin/bash
# RT_PROFILE section shares of a profile build (tools/ab_bench.py arm `prof`, or `prof_split`: the random draws booked apart from
# lane_shade) on the headline frame, the cover frame at spp 32 through the unlit and the lit kernel, the lit cover frame and cfg1.
cd "${GRAFT_REPO_ROOT:-.}"
LIB=${1:-build/ab/librt_hip_prof.so}
{
echo "# $LIB (tools/diag.py --lib): shader-clock cycles per wave iteration by kernel section"
for S in "" "--scene build/ab/cover_spp32.json" "--scene build/ab/cover_spp32.json --opt force_lit=1" "--scene build/ab/lit_cover_spp32.json" "--scene scenes/cfg1_test_800x600_spp16.json"; do
  echo "## diag.py $S"; timeout 100 python tools/diag.py --lib $LIB $S --reps 4 2>/dev/null | tail -1
done
} | tee gpurun_out/sections_$(basename $LIB .so).log
