#!/bin/bash
# round-2 experiment run 6: deeper-ring tile candidates + tile-walk stage of the in-situ tuner, big-tile K-split rule, LayerNorm rows per wave
set -u
OUT=gpurun_out/r02_run6; mkdir -p $OUT
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q -k "not sdxl and not real" 2>&1 | tail -8 | tee $OUT/pytest_kernels.txt
echo "== LN rows per wave"; timeout 200 python scripts/bench_norm.py --ln-only 2>&1 | grep -v amdgpu.ids | tee $OUT/ln_rpw.txt
echo "== forced K-split A/B"
run() { SPLIT=$3 timeout 120 python scripts/one_igemm.py $1 $2 30 2>&1 | grep -v amdgpu.ids; }
( for sh in conv_l2_1280 lin_m4096_ffout; do run $sh 7 0; run $sh 8 0; run $sh 8 2; run $sh 5 4; run $sh 4 2; done
  run lin_m4096_1280 7 0; run lin_m4096_1280 8 2
  for sh in conv_m2048_1280; do run $sh 1 0; run $sh 7 0; run $sh 8 4; run $sh 8 2; run $sh 5 8; done
  run lin_m2048_ffout 1 0; run lin_m2048_ffout 7 0; run lin_m2048_ffout 8 2; run lin_m2048_ffout 8 4
  run conv_m8192_640 7 0; run conv_m8192_640 8 0; run conv_m8192_640 8 2 ) | tee $OUT/split_ab.txt
echo "== in-situ A/B sd15 rows 16"
prof() { TUNE_MASK=$3 BIG_SPLIT=$4 LN_RPW=$5 timeout 300 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$6.txt 2>&1; grep "^# " $OUT/prof_$1_$6.txt; }
prof sd15 16 0x5f2 0 1 base
prof sd15 16 0xffffffff 0 1 tune
prof sd15 16 0xffffffff 32 0 all
prof sd15 16 0x5f2 32 0 split_ln
echo "== in-situ A/B sdxl rows 4"
prof sdxl 4 0x5f2 0 1 base
prof sdxl 4 0xffffffff 32 0 all
echo "== in-situ sdxl rows 2 (edit)"
prof sdxl 2 0x5f2 0 1 base
prof sdxl 2 0xffffffff 32 0 all
du -sh $OUT
