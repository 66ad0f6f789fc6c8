#!/bin/bash
# Round-4 call 5: tile32_kernel (32-deep K-tiles for every activation map; 256 x 256 / 256 x 320 on four stages): kernel tests,
# same-box forward A/Bs, model-level parity with the tuner offering them
set -u
OUT=gpurun_out/r04_call5; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
echo "== kernel tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.txt 2>&1; tail -6 $OUT/pytest_kernels.txt | cut -c1-600
OLD=0xffec5fff     # tuner mask without 13 / 15 / 16 / 17 / 20
LIN=0xffefdfff     # with 15 / 16 / 17 (now also for convolutions), without 13 / 20
echo "== A/B sd15 b8"; timeout 900 python scripts/r04_runs/ab_forward.py sd15 8 "none:mask=$OLD;t32_3stage:mask=$LIN;t32_all:mask=0xffffffff" --table > $OUT/ab_sd15_b8.txt 2>&1; head -5 $OUT/ab_sd15_b8.txt | cut -c1-420
echo "== A/B sdxl b2"; timeout 1500 python scripts/r04_runs/ab_forward.py sdxl 2 "none:mask=$OLD;t32_3stage:mask=$LIN;t32_all:mask=0xffffffff" --table > $OUT/ab_sdxl_b2.txt 2>&1; head -5 $OUT/ab_sdxl_b2.txt | cut -c1-420
echo "== table sd15"; sed -n 5,45p $OUT/ab_sd15_b8.txt
echo "== model-level parity"
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py -x -q -m gpu > $OUT/pytest_unet_vae.txt 2>&1; tail -3 $OUT/pytest_unet_vae.txt
echo "== VAE"; timeout 300 python scripts/profile_vae.py 8 64 > $OUT/vae_b8_64.txt 2>&1; head -14 $OUT/vae_b8_64.txt
