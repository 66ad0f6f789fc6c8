#!/bin/bash
# Round-5 call 11: prestats GroupNorm restricted to >= 32x32 maps - kernel test + same-box A/B
set -u
OUT=gpurun_out/r05_call11; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.txt 2>&1; tail -3 $OUT/pytest_kernels.txt | cut -c1-600
V="own_pass:prestats=0;prestats:prestats=1;own_pass2:prestats=0;prestats2:prestats=1"
echo "== A/B sd15 b8"; timeout 900 python scripts/r05_runs/ab_forward.py sd15 8 "$V" > $OUT/ab_sd15_b8.txt 2>&1; head -7 $OUT/ab_sd15_b8.txt | cut -c1-330
echo "== A/B sdxl b2"; timeout 1500 python scripts/r05_runs/ab_forward.py sdxl 2 "$V" > $OUT/ab_sdxl_b2.txt 2>&1; head -7 $OUT/ab_sdxl_b2.txt | cut -c1-330
