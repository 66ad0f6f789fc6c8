#!/bin/bash
# Round-5 call 15: XCD-blocked 2-D tile walk (IGemmArgs::walk_bn, walk_plan): kernel tests, tuning invariance, same-box A/B
# (outputs must be bit-identical between the variants: rel-L2 0.00e+00), per-launch table
set -u
OUT=gpurun_out/r05_call15; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.txt 2>&1; tail -3 $OUT/pytest_kernels.txt | cut -c1-600
echo "== tuning invariance"; timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "tile_tuning or tuning_on_and_off or deterministic" > $OUT/pytest_unet_tuning.txt 2>&1; tail -3 $OUT/pytest_unet_tuning.txt | cut -c1-400
V="walk1d:blocked=0,mask=0xf1ffffff;blocked:blocked=1,mask=0xf1ffffff;walk1d_b:blocked=0,mask=0xf1ffffff;blocked_b:blocked=1,mask=0xf1ffffff"
echo "== A/B sdxl b2"; timeout 1500 python scripts/r05_runs/ab_forward.py sdxl 2 "$V" --table > $OUT/ab_sdxl_b2.txt 2>&1; head -7 $OUT/ab_sdxl_b2.txt | cut -c1-330; grep "geglu" $OUT/ab_sdxl_b2.txt | head -4
echo "== A/B sd15 b8"; timeout 900 python scripts/r05_runs/ab_forward.py sd15 8 "$V" --table > $OUT/ab_sd15_b8.txt 2>&1; head -7 $OUT/ab_sd15_b8.txt | cut -c1-330; grep "geglu" $OUT/ab_sd15_b8.txt | head -4
