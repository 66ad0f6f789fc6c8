#!/bin/bash
# Round-5 call 10: GroupNorm statistics from the producers' epilogues (IGemmArgs::gstat + cfgpp_op_groupnorm_pre): kernel tests,
# model-level parity / determinism / tuning invariance, same-box A/B of whole forwards (switch: cfgpp_groupnorm_set_prestats), VAE
set -u
OUT=gpurun_out/r05_call10; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.txt 2>&1; tail -12 $OUT/pytest_kernels.txt | cut -c1-1500
echo "== model-level"; timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py -x -q -m gpu > $OUT/pytest_unet_vae.txt 2>&1; tail -6 $OUT/pytest_unet_vae.txt | cut -c1-600
V="own_pass:prestats=0;prestats:prestats=1"
echo "== A/B sd15 b8"; timeout 900 python scripts/r05_runs/ab_forward.py sd15 8 "$V" --table > $OUT/ab_sd15_b8.txt 2>&1; head -5 $OUT/ab_sd15_b8.txt | cut -c1-520; grep "groupnorm" $OUT/ab_sd15_b8.txt | head -12
echo "== A/B sdxl b2"; timeout 1500 python scripts/r05_runs/ab_forward.py sdxl 2 "$V" --table > $OUT/ab_sdxl_b2.txt 2>&1; head -5 $OUT/ab_sdxl_b2.txt | cut -c1-520; grep "groupnorm" $OUT/ab_sdxl_b2.txt | head -10
echo "== VAE"; CFGPP_GN_PRESTATS=0 timeout 300 python scripts/profile_vae.py 8 64 > $OUT/vae_b8_64_own_pass.txt 2>&1; timeout 300 python scripts/profile_vae.py 8 64 > $OUT/vae_b8_64_prestats.txt 2>&1; head -3 $OUT/vae_b8_64_own_pass.txt; grep groupnorm $OUT/vae_b8_64_own_pass.txt | head -4; tail -n 1 $OUT/vae_b8_64_own_pass.txt; head -3 $OUT/vae_b8_64_prestats.txt; grep groupnorm $OUT/vae_b8_64_prestats.txt | head -4; tail -n 1 $OUT/vae_b8_64_prestats.txt
