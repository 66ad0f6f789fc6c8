#!/bin/bash
# Round-4 call 9: attention with four workgroups per CU (2-stage ring, <= 128 VGPRs) vs three: kernel alone, then whole forwards
set -u
OUT=gpurun_out/r04_call9; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
timeout 600 python scripts/r04_runs/ab_attention.py > $OUT/attention_occupancy_alone.txt 2>&1; grep -v amdgpu.ids $OUT/attention_occupancy_alone.txt
timeout 900 python scripts/r04_runs/ab_forward.py sd15 8 "occ3:attnocc=3;occ4:attnocc=4;occ3b:attnocc=3;occ4b:attnocc=4" > $OUT/ab_sd15_b8.txt 2>&1; grep -v amdgpu.ids $OUT/ab_sd15_b8.txt | cut -c1-330
timeout 1200 python scripts/r04_runs/ab_forward.py sdxl 2 "occ3:attnocc=3;occ4:attnocc=4" > $OUT/ab_sdxl_b2.txt 2>&1; grep -v amdgpu.ids $OUT/ab_sdxl_b2.txt | cut -c1-330
