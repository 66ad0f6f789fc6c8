#!/bin/bash
# Round 3, GPU call 5: epilogue parameters through LDS (LDS-DMA at kernel start) + residual loads in one block: kernel tests,
# timelines of the launches whose epilogues were 5-22 us, in-situ profiles (LayerNorm fusion off / on).
set -u
OUT=gpurun_out/r03_call5; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 tests"
timeout 500 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_kernels.txt
timeout 500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_unet_vae.txt
echo "== 2 timelines"
timeout 300 python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" "geglu HW=1024 N=10240 K=1280" "heads HW=1024 N=3840 K=1280" \
    "conv3x3 amode=1 HW=16384 N=320 K=2880 +res" > $OUT/timeline_sdxl_rows4.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_sdxl_rows4.txt | grep -E "^##|prologue \(|per K-tile|epilogue|workgroup total|span"
timeout 300 python scripts/igemm_timeline.py sd15 16 "geglu HW=4096 N=2560 K=320" "linear HW=4096 N=320 K=320 +res" "conv3x3 amode=1 HW=4096 N=320 K=2880 +res" \
    "conv3x3 amode=1 HW=256 N=1280 K=11520 +res" "heads HW=4096 N=960 K=320" > $OUT/timeline_sd15_rows16.txt 2>&1; grep -v amdgpu.ids $OUT/timeline_sd15_rows16.txt | grep -E "^##|prologue \(|per K-tile|epilogue|workgroup total|span"
echo "== 3 in-situ"
prof() { name=$1; shift; env "$@" timeout 200 python scripts/profile_unet.py ${CFG} > $OUT/prof_${CFGN}_$name.txt 2>&1; echo "$name: $(grep '^# ' $OUT/prof_${CFGN}_$name.txt | head -2 | tr '\n' ' ')"; }
for c in "sd15 16" "sdxl 4"; do
  CFG="$c"; CFGN=$(echo $c | tr ' ' '_')
  echo "-- $c"
  prof base FUSE_LN=0
  prof ln FUSE_LN=1
done
du -sh $OUT
