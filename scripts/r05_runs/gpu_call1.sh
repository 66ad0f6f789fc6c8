#!/bin/bash
# Round-5 call 1: big4_kernel (one wave per SIMD: 256 x 256 / 128 x 320 / 128 x 256 on four waves, configs 24 / 25 / 26): kernel tests,
# same-box forward A/Bs with and without the new tuner candidates, per-workgroup timelines of the launches VERDICT r04 names
set -u
OUT=gpurun_out/r05_call1; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
git rev-parse HEAD > $OUT/HEAD.txt 2>/dev/null
echo "== kernel tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.txt 2>&1; tail -6 $OUT/pytest_kernels.txt | cut -c1-800
echo "== tuning invariance + pin cache"
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "tile_tuning or tuning_on_and_off or pins_persist or deterministic" > $OUT/pytest_unet_tuning.txt 2>&1; tail -4 $OUT/pytest_unet_tuning.txt | cut -c1-600
OLD=0xf8ffffff     # tuner mask without 24 / 25 / 26
echo "== A/B sd15 b8"; timeout 900 python scripts/r05_runs/ab_forward.py sd15 8 "base:mask=$OLD;big4:mask=0xffffffff" --table > $OUT/ab_sd15_b8.txt 2>&1; head -4 $OUT/ab_sd15_b8.txt | cut -c1-520
echo "== A/B sdxl b2"; timeout 1500 python scripts/r05_runs/ab_forward.py sdxl 2 "base:mask=$OLD;big4:mask=0xffffffff" --table > $OUT/ab_sdxl_b2.txt 2>&1; head -4 $OUT/ab_sdxl_b2.txt | cut -c1-520
echo "== table sd15 (big4 variant)"; sed -n 4,50p $OUT/ab_sd15_b8.txt
echo "== table sdxl (big4 variant)"; sed -n 4,40p $OUT/ab_sdxl_b2.txt
echo "== timelines sd15 rows 16"
timeout 600 python scripts/igemm_timeline.py sd15 16 "conv3x3 amode=1 HW=4096 N=320 K=2880 +res" "geglu HW=4096 N=2560 K=320" "heads HW=4096 N=960 K=320" "conv3x3 amode=1 HW=1024 N=640 K=5760 +res" "linear HW=4096 N=320 K=320 +res" > $OUT/timeline_sd15.txt 2>&1; cat $OUT/timeline_sd15.txt | cut -c1-300 | head -80
echo "== timelines sdxl rows 4"
timeout 900 python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" "heads HW=1024 N=1280 K=1280" "geglu HW=1024 N=10240 K=1280" "linear HW=4096 N=640 K=640 +res" > $OUT/timeline_sdxl.txt 2>&1; cat $OUT/timeline_sdxl.txt | cut -c1-300 | head -70
