#!/bin/bash
# Round 3, GPU call 7: the whole GPU suite on the build that ships (heads epilogues without per-piece integer divisions,
# reciprocal-multiply index math, run-time time-embedding segments, no par_late), then the QKV-projection timeline (epilogue
# cycles before: 34 000 per workgroup on the 256 x 256 tile), per-CU workgroup turnover, and the in-situ profiles.
set -u
OUT=gpurun_out/r03_call7; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 whole GPU suite"
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== 2 timelines"
timeout 300 python scripts/igemm_timeline.py sdxl 4 "heads HW=1024 N=3840 K=1280" "heads HW=1024 N=1280 K=1280" "geglu HW=1024 N=10240 K=1280" > $OUT/timeline_sdxl_rows4.txt 2>&1
grep -v amdgpu.ids $OUT/timeline_sdxl_rows4.txt | grep -E "^##|tile id|entry -> first DMA|prologue \(|per K-tile|epilogue|workgroup total|span|CUs seen|exit -> next"
timeout 300 python scripts/igemm_timeline.py sd15 16 "geglu HW=4096 N=2560 K=320" "heads HW=4096 N=960 K=320" "linear HW=4096 N=320 K=1280 +res" > $OUT/timeline_sd15_rows16.txt 2>&1
grep -v amdgpu.ids $OUT/timeline_sd15_rows16.txt | grep -E "^##|tile id|entry -> first DMA|prologue \(|per K-tile|epilogue|workgroup total|span|CUs seen|exit -> next"
echo "== 3 in-situ"
for c in "sd15 16" "sdxl 4"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 python scripts/profile_unet.py $c > $OUT/prof_$n.txt 2>&1; echo "$c: $(grep '^# ' $OUT/prof_$n.txt | head -2 | tr '\n' ' ')"
done
du -sh $OUT
