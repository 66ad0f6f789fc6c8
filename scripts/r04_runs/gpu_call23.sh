#!/bin/bash
# call 23: SD1.5 b8 forwards, attention kernel switched in one engine, with per-family profiled sums
set -u
O=gpurun_out/r04_call23; mkdir -p $O
timeout 400 python scripts/r04_runs/ab_forward_attn.py sd15 8 1,2 2>&1 | grep -v amdgpu.ids | tee $O/forward_attention_modes_sd15_b8.txt
