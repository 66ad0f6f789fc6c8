#!/bin/bash
# Round-4 call 10: attention with four workgroups per CU as the default: the attention tests (UNet sizes, rescale branch, kernels), then
# both occupancies once more alone and inside forwards on this build
set -u
OUT=gpurun_out/r04_call10; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -q -m gpu -k "attention or attn" > $OUT/pytest_attention.txt 2>&1; tail -4 $OUT/pytest_attention.txt
timeout 600 python scripts/r04_runs/ab_attention.py > $OUT/attention_occupancy_alone.txt 2>&1; grep -v amdgpu.ids $OUT/attention_occupancy_alone.txt
timeout 900 python scripts/r04_runs/ab_forward.py sd15 8 "occ3:attnocc=3;occ4:attnocc=4" > $OUT/ab_sd15_b8.txt 2>&1; grep -v amdgpu.ids $OUT/ab_sd15_b8.txt | cut -c1-330
