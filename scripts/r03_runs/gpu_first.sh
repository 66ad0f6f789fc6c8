#!/bin/bash
# First GPU run of round 3 (prepared at the end of round 2, when the GPU budget was spent):
#   1. first-contact tests of the code written without hardware (text tower on the HIP kernels; head-major epilogue of the
#      16x16x32 tile) - opt-in guards on
#   2. in-situ A/Bs that need no new code: the 16x16x32 tile on grids of two full rounds (M = 16384 x N = 640), its heads epilogue
#   3. per-launch profile of the VAE decode (never profiled per launch: 18 % of the Lightning job)
# Expected: ~6 GPU-minutes.
set -u
OUT=gpurun_out/r03_first; mkdir -p $OUT
echo "== first-contact tests"
CFGPP_TEST_TEXT=1 CFGPP_TEST_MF16_HEADS=1 timeout 300 python -m pytest tests/test_gpu_text.py tests/test_gpu_kernels.py -m gpu -q -k "text or head_dim_40" 2>&1 | tail -25 | tee $OUT/pytest_first_contact.txt
prof() { MF16_ROUNDS=$3 MF16_HEADS=$4 timeout 120 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$2_r$3_h$4.txt 2>&1; grep "^# " $OUT/prof_$1_$2_r$3_h$4.txt | head -2; }
echo "== sd15 rows 16: rule as shipped | + two-round grids | + heads"
prof sd15 16 1 0
prof sd15 16 2 0
prof sd15 16 2 1
echo "== sdxl rows 4: rule as shipped | + two-round grids | + heads (N = 3840: three rounds -> MF16_ROUNDS=3)"
prof sdxl 4 1 0
prof sdxl 4 3 1
echo "== VAE decode per launch"
timeout 120 python scripts/profile_vae.py 8 64 > $OUT/vae_b8_64.txt 2>&1; head -14 $OUT/vae_b8_64.txt
timeout 180 python scripts/profile_vae.py 8 128 > $OUT/vae_b8_128.txt 2>&1; head -14 $OUT/vae_b8_128.txt
du -sh $OUT
