#!/bin/bash
# Round 3, GPU call 1: everything that had never run / never been compared, plus the first measurements of the round.
#   1. first-contact tests (HIP text tower, head-major epilogue of the 16x16x32 tile) with the opt-in guards ON
#   2. pin of the "cuda" scalar semantics against torch-ROCm; step-kernel tests after the division change
#   3. parity at the C4 / C5 plan sizes (SDXL 2 / 4 / 16 rows @128^2) and the VAE at 1024^2
#   4. per-workgroup timelines of representative igemm launches inside real forwards
#   5. in-situ A/Bs that need no new code (prepared at the end of round 2) + per-launch VAE profiles
set -u
OUT=gpurun_out/r03_call1; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 first contact"
CFGPP_TEST_TEXT=1 CFGPP_TEST_MF16_HEADS=1 timeout 400 python -m pytest tests/test_gpu_text.py tests/test_gpu_kernels.py -m gpu -q -x -k "text or head_dim_40" 2>&1 | tail -30 | tee $OUT/pytest_first_contact.txt
CFGPP_TEST_TEXT=1 timeout 300 python -m pytest tests/test_gpu_text.py -m gpu -q 2>&1 | tail -30 | tee $OUT/pytest_text_all.txt
echo "== 2 cuda semantics pin + step kernels"
timeout 300 python -m pytest tests/test_gpu_torch_semantics.py tests/test_gpu_step.py -m gpu -q 2>&1 | tail -40 | tee $OUT/pytest_semantics.txt
echo "== 3 plan-size parity"
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "every_bench_plan_size or at_1024" 2>&1 | tail -30 | tee $OUT/pytest_plan_sizes.txt
echo "== 4 timelines"
timeout 300 python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" "heads HW=1024 N=1280 K=1280" "heads HW=1024 N=3840 K=1280" \
    "linear HW=1024 N=1280 K=5120 +res" "geglu HW=1024 N=10240 K=1280" "conv3x3 amode=1 HW=1024 N=1280 K=11520 +res" "linear HW=4096 N=640 K=640 +res" \
    "conv3x3 amode=1 HW=16384 N=320 K=2880 +res" > $OUT/timeline_sdxl_rows4.txt 2>&1; tail -n +1 $OUT/timeline_sdxl_rows4.txt | grep -v amdgpu.ids | head -120
timeout 300 python scripts/igemm_timeline.py sd15 16 "geglu HW=4096 N=2560 K=320" "heads HW=4096 N=960 K=320" "linear HW=4096 N=320 K=320 +res" \
    "conv3x3 amode=1 HW=4096 N=320 K=2880 +res" "conv3x3 amode=1 HW=256 N=1280 K=11520 +res" "conv3x3 amode=1 HW=64 N=1280 K=11520 +res" \
    "linear HW=256 N=1280 K=1280 +res" "geglu HW=1024 N=5120 K=640" "conv3x3 amode=1 HW=1024 N=640 K=5760 +res" > $OUT/timeline_sd15_rows16.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_sd15_rows16.txt | head -130
echo "== 5 in-situ A/Bs + VAE profiles"
prof() { MF16_ROUNDS=$3 MF16_HEADS=$4 timeout 150 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$2_r$3_h$4.txt 2>&1; grep "^# " $OUT/prof_$1_$2_r$3_h$4.txt | head -2; }
prof sd15 16 1 0
prof sd15 16 2 0
prof sd15 16 2 1
prof sdxl 4 1 0
prof sdxl 4 3 1
timeout 150 python scripts/profile_vae.py 8 64 > $OUT/vae_b8_64.txt 2>&1; head -16 $OUT/vae_b8_64.txt
timeout 200 python scripts/profile_vae.py 8 128 > $OUT/vae_b8_128.txt 2>&1; head -16 $OUT/vae_b8_128.txt
cp gpurun_out/parity_r03.jsonl $OUT/ 2>/dev/null
du -sh $OUT
