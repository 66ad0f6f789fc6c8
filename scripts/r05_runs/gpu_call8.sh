#!/bin/bash
# Round-5 call 8: config 27 (128 x 320 on four waves, 32-deep two-stage ring, TWO workgroups per CU = config 5's tile as two
# independent halves): kernel tests, forced-hint per-launch profile, same-box forward A/B
set -u
OUT=gpurun_out/r05_call8; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tile or big or heads or gemm or conv" > $OUT/pytest_kernels.txt 2>&1; tail -3 $OUT/pytest_kernels.txt | cut -c1-600
echo "== forced hint 27, sd15 b8"; timeout 400 python scripts/r05_runs/force_hint_profile.py sd15 8 27 > $OUT/force27_sd15.txt 2>&1; cat $OUT/force27_sd15.txt | cut -c1-200
echo "== forced hint 27, sdxl b2"; timeout 600 python scripts/r05_runs/force_hint_profile.py sdxl 2 27 > $OUT/force27_sdxl.txt 2>&1; cat $OUT/force27_sdxl.txt | cut -c1-200
V="base:mask=0xf1ffffff;c27:mask=0xf9ffffff"
echo "== A/B sd15 b8"; timeout 900 python scripts/r05_runs/ab_forward.py sd15 8 "$V" > $OUT/ab_sd15_b8.txt 2>&1; head -5 $OUT/ab_sd15_b8.txt | cut -c1-520
echo "== A/B sdxl b2"; timeout 1500 python scripts/r05_runs/ab_forward.py sdxl 2 "$V" > $OUT/ab_sdxl_b2.txt 2>&1; head -5 $OUT/ab_sdxl_b2.txt | cut -c1-520
