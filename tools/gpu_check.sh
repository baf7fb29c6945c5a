#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench line, rocprofv3 kernel stats, PMC passes (+ optional A/B,
# + optional full-size runs of every BASELINE config and the one-rank-at-a-time shard emulation).
# Usage (from the build container):  gpurun --timeout 900 -- 'bash tools/gpu_check.sh [ab] [pmc] [configs]'
# Every step runs under `timeout` and nothing reads stdin: a stuck step must not eat GPU budget.
set -u
exec </dev/null
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/summary.txt
T0=$(date +%s)
stamp() { echo "$1 rc=$2 t=$(( $(date +%s) - T0 ))s" | tee -a $OUT/summary.txt; }
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device.txt; nproc >> $OUT/device.txt
timeout 240 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; stamp smoke $?
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; stamp pytest $?
tail -4 $OUT/pytest_gpu.log
if [[ " $* " == *" ab "* ]]; then
  timeout 300 python tools/ab_bench.py run --rounds 5 > $OUT/ab.log 2>&1; stamp ab $?
  grep -v amdgpu.ids $OUT/ab.log | tail -20
fi
export TMPDIR=/tmp
REPO=$PWD
if [[ " $* " == *" pmc "* ]]; then
  # HBM traffic: separate PMC passes, counters only (no trace domains), per the microarch guide
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d "$REPO/$OUT/pmc_$C" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs ) > $OUT/pmc_$C.log 2>&1; stamp pmc_$C $?
  done
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$REPO/$OUT/pmc_sq" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs ) > $OUT/pmc_sq.log 2>&1; stamp pmc_sq $?
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$REPO/$OUT/pmc_sq2" -o bench -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs ) > $OUT/pmc_sq2.log 2>&1; stamp pmc_sq2 $?
  python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json 2>$OUT/pmc_summary.err; stamp pmc_summary $?
  cat $OUT/pmc_summary.json
fi
if [[ " $* " == *" pmcs "* ]]; then
  # counters of the configs where the memory system takes part: textures (cfg3), tables in L2 (cfg5), the lit test scene (cfg1)
  timeout 2400 bash tools/pmc_scene.sh "${RT_TAG:-rXX}" cfg1=scenes/cfg1_test_800x600_spp16.json cfg3=scenes/cfg3_cover_4k_textured.json cfg4=scenes/cfg4_cover_4k_textured_spp512.json cfg5=procedural:50:2048 litcover=build/ab/lit_cover_spp32.json > $OUT/pmcs.log 2>&1; stamp pmcs $?
  grep -E "^cfg[0-9] (\{|pass)" $OUT/pmcs.log | cut -c1-400
fi
if [[ " $* " == *" group "* ]]; then
  # the product's multi-GPU path in ONE process (bench.py --gpus N without torchrun), ranks sharing this box's GPU
  RT_GPUS_EMULATE=1 timeout 300 python bench.py --gpus 4 --steps 5 --warmup 2 > $OUT/bench_group4.log 2>$OUT/bench_group4.err; stamp bench_group4 $?
  tail -1 $OUT/bench_group4.log | cut -c1-1500
  RT_GPUS_EMULATE=1 timeout 300 python bench.py --gpus 8 --steps 5 --warmup 2 > $OUT/bench_group8.log 2>$OUT/bench_group8.err; stamp bench_group8 $?
  tail -1 $OUT/bench_group8.log | cut -c1-1500
  timeout 120 python bench.py --gpus 8 --steps 2 --warmup 1 > $OUT/bench_group8_refused.log 2>/dev/null; stamp bench_group8_refused $?
  tail -1 $OUT/bench_group8_refused.log | cut -c1-400
fi
if [[ " $* " == *" configs "* ]]; then
  {
    echo "# full-size runs of every BASELINE config, current kernel"
    for S in scenes/cfg1_test_800x600_spp16.json scenes/cfg3_cover_4k_textured.json scenes/cfg4_cover_4k_textured_spp512.json; do
      timeout 120 python tools/diag.py --scene $S --reps 2 2>/dev/null | tail -1
    done
    timeout 200 python tools/diag.py --procedural 50 --spp 2048 --reps 1 2>/dev/null | tail -1
    echo "# one rank's shard of the headline frame at a time (2-scanline interleave): G = 2, 4, 8"
    for SH in 0,2,2 1,2,2 0,4,2 3,4,2 0,8,2 3,8,2 7,8,2; do
      echo -n "shard $SH "; timeout 60 python tools/diag.py --shard $SH --reps 10 2>/dev/null | tail -1 | cut -c 50-140
    done
    echo "# queue order (tile_order 0 = top row first, 1 = bottom row first, 2 = deepest tiles of the previous frame first): 1/8 shards, whole frame"
    for O in 0 1 2; do
      for SH in 0,8,2 3,8,2 7,8,2; do
        echo -n "order $O shard $SH "; timeout 60 python tools/diag.py --shard $SH --reps 10 --opt tile_order=$O 2>/dev/null | tail -1 | cut -c 50-140
      done
      echo -n "order $O whole frame "; timeout 60 python tools/diag.py --reps 6 --opt tile_order=$O 2>/dev/null | tail -1 | cut -c 50-140
    done
    if [ -f build/ab/librt_hip_prof.so ]; then
      echo "# RT_PROFILE build: section shares, wave timeline (full frame, then rank 3 of 8)"
      timeout 60 python tools/diag.py --lib build/ab/librt_hip_prof.so --reps 5 2>/dev/null | tail -1
      timeout 60 python tools/diag.py --lib build/ab/librt_hip_prof.so --shard 3,8,2 --reps 5 2>/dev/null | tail -1
    fi
  } > $OUT/all_configs.log 2>&1; stamp configs $?
  cut -c1-160 $OUT/all_configs.log
fi
if [[ " $* " == *" sweep "* ]]; then
  {
    echo "# tile shape (0 = squares, 1 = scanline runs, 2 = 16x4, 3 = 32x2): whole frame, 1/8 shard, cfg3 4K"
    for SH in 0 2 3 1; do
      echo -n "tile_shape=$SH whole "; timeout 60 python tools/diag.py --reps 6 --opt tile_shape=$SH 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
      echo -n "tile_shape=$SH shard 3/8 "; timeout 60 python tools/diag.py --shard 3,8,2 --reps 8 --opt tile_shape=$SH 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
      echo -n "tile_shape=$SH cfg3 "; timeout 60 python tools/diag.py --scene scenes/cfg3_cover_4k_textured.json --reps 2 --opt tile_shape=$SH 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
    done
    echo "# chunk_spp x tile_log2 on the whole headline frame (kernel_ms, best of 6; tile_order feedback on)"
    for TL in 1 2 3; do for CS in 4 8 16 32 64; do
      echo -n "tile_log2=$TL chunk_spp=$CS "; timeout 60 python tools/diag.py --reps 6 --opt tile_log2=$TL chunk_spp=$CS 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
    done; done
    echo "# 1/8 shard (rank 3)"
    for TL in 0 1 2; do for CS in 8 16 32 64 128; do
      echo -n "tile_log2=$TL chunk_spp=$CS "; timeout 60 python tools/diag.py --shard 3,8,2 --reps 8 --opt tile_log2=$TL chunk_spp=$CS 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])"
    done; done
  } > $OUT/sweep.log 2>&1; stamp sweep $?
  cat $OUT/sweep.log
fi
# the bench line last (its roofline reads the newest profiles/rNN_*pmc.json: copy gpurun_out/pmc_summary.json there after a pmc run)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/prof" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs ) > $OUT/rocprof.log 2>&1
stamp rocprof $?
STATS=$(find $OUT/prof -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp "$STATS" $OUT/kernel_stats.csv; head -6 "$STATS"; fi
timeout 400 python bench.py --steps 10 --warmup 2 > $OUT/bench.log 2>&1; stamp bench $?
tail -1 $OUT/bench.log
echo done | tee -a $OUT/summary.txt
