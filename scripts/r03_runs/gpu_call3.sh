#!/bin/bash
# Round 3, GPU call 3: the new igemm16 K-tile schedule (mid-tile barrier, fragment reads half a tile ahead, staggered LDS-DMA
# issue) - race screen, timelines, in-situ profiles - plus the whole GPU suite on this build (guards dropped, device kernargs).
set -u
OUT=gpurun_out/r03_call3; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 targeted tests"
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_torch_semantics.py -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_targeted.txt
echo "== 2 timelines (igemm16 launches)"
timeout 300 python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" "linear HW=1024 N=1280 K=5120 +res" "conv3x3 amode=1 HW=1024 N=1280 K=11520 +res" \
    "heads HW=1024 N=1280 K=1280" > $OUT/timeline_sdxl_rows4.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_sdxl_rows4.txt | head -70
echo "== 3 in-situ profiles"
for c in "sd15 16" "sdxl 4"; do
  set -- $c
  timeout 200 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$2.txt 2>&1; grep "^# " $OUT/prof_$1_$2.txt | head -2
  MF16_ROUNDS=2 timeout 200 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$2_rounds2.txt 2>&1; grep "^# " $OUT/prof_$1_$2_rounds2.txt | head -2
done
echo "== 4 whole GPU suite"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
cp gpurun_out/parity_r03.jsonl $OUT/ 2>/dev/null
du -sh $OUT
