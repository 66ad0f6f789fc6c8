#!/bin/bash
# call 16: the woven attention kernel (attn64w_kernel) against the flash loop, alone
set -u
O=gpurun_out/r04_call16; mkdir -p $O
timeout 300 python scripts/r04_runs/ab_attention_woven.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_woven_alone.txt
