#!/bin/bash
# Round-5 call 9 (final build): the kernel / tuning / VAE / SDXL-chain tests the last commits touch, then gpu_final.sh all
# (PMC + kernel-trace passes for the four bench populations, the bench lines, rocprofv3 --stats of the bench command, per-launch tables)
set -u
OUT=gpurun_out/r05_call9; mkdir -p $OUT profiles/r05
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
git rev-parse HEAD > profiles/r05/HEAD.txt 2>/dev/null; cp profiles/r05/HEAD.txt $OUT/
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vae.py tests/test_gpu_step.py tests/test_gpu_unet.py -q -m gpu -p no:cacheprovider -k "not vs_oracle" > $OUT/pytest_subset.txt 2>&1; tail -3 $OUT/pytest_subset.txt | cut -c1-400
bash scripts/r05_runs/gpu_final.sh all
