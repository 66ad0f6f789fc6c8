#!/bin/bash
# call 20: whole forwards with the attention kernel switched in one engine: SD1.5 b8, SDXL b2
set -u
O=gpurun_out/r04_call20; mkdir -p $O
timeout 400 python scripts/r04_runs/ab_forward_attn.py sd15 8 2>&1 | grep -v amdgpu.ids | tee $O/forward_attention_modes_sd15_b8.txt
timeout 500 python scripts/r04_runs/ab_forward_attn.py sdxl 2 2>&1 | grep -v amdgpu.ids | tee $O/forward_attention_modes_sdxl_b2.txt
