#!/bin/bash
# Round 3, GPU call 2: first contact (mf16 heads, text tower after the key fix), device-scalar diagnosis, kernel-argument
# placement A/B (HIP_FORCE_DEV_KERNARG), prologue sub-stamps in the timelines, residual-prefetch epilogue timelines.
set -u
OUT=gpurun_out/r03_call2; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 first contact"
CFGPP_TEST_MF16_HEADS=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "head_dim_40 or heads_projection" 2>&1 | tail -15 | tee $OUT/pytest_mf16_heads.txt
CFGPP_TEST_TEXT=1 timeout 300 python -m pytest tests/test_gpu_text.py -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_text.txt
echo "== 2 device scalar rules"
timeout 120 python scripts/r03_runs/diag_device_scalar.py 2>&1 | grep -v amdgpu.ids | tee $OUT/diag_device_scalar.txt
echo "== 3 kernel-argument placement"
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 200 python scripts/profile_unet.py sd15 16 > $OUT/prof_sd15_kernarg$v.txt 2>&1; echo "HIP_FORCE_DEV_KERNARG=$v sd15:"; grep "^# " $OUT/prof_sd15_kernarg$v.txt | head -2
done
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 200 python scripts/profile_unet.py sdxl 4 > $OUT/prof_sdxl_kernarg$v.txt 2>&1; echo "HIP_FORCE_DEV_KERNARG=$v sdxl:"; grep "^# " $OUT/prof_sdxl_kernarg$v.txt | head -2
done
echo "== 4 timelines with prologue sub-stamps (this build: residual requested before the LDS transpose)"
timeout 300 python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" "heads HW=1024 N=3840 K=1280" "geglu HW=1024 N=10240 K=1280" \
    "linear HW=4096 N=640 K=640 +res" > $OUT/timeline_sdxl_rows4.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_sdxl_rows4.txt | head -80
HIP_FORCE_DEV_KERNARG=1 timeout 300 python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" "geglu HW=1024 N=10240 K=1280" > $OUT/timeline_sdxl_rows4_kernarg1.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_sdxl_rows4_kernarg1.txt | head -40
timeout 300 python scripts/igemm_timeline.py sd15 16 "geglu HW=4096 N=2560 K=320" "linear HW=4096 N=320 K=320 +res" "conv3x3 amode=1 HW=4096 N=320 K=2880 +res" \
    "linear HW=256 N=1280 K=1280 +res" > $OUT/timeline_sd15_rows16.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_sd15_rows16.txt | head -80
du -sh $OUT
