#!/bin/bash
# closing shot of round 2: correctness of the 16x16x32-MFMA tile (forced configs 18 / 19), then the rule in situ
set -u
OUT=gpurun_out/r02_mf16; mkdir -p $OUT
timeout 75 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "big" 2>&1 | tail -6 | tee $OUT/pytest_mf16.txt
prof() { MF16=$3 timeout 45 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$2_mf$3.txt 2>&1; grep "^# " $OUT/prof_$1_$2_mf$3.txt | head -2; grep -E "HW=256 N=1280 K=(11520|5120|1280) \+res" $OUT/prof_$1_$2_mf$3.txt; }
prof sd15 16 3
prof sd15 16 0
prof sd15 16 4
