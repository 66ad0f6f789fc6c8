#!/bin/bash
# call 18: woven attention kernel v2 (4-stage rings, immediate LDS offsets, counted lgkm waits, alternating QK^T chains): A/B + switched-off timings
set -u
O=gpurun_out/r04_call18; mkdir -p $O
timeout 300 python scripts/r04_runs/ab_attention_woven.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_woven_alone.txt
timeout 300 python scripts/r04_runs/diag_attention_woven.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_woven_diag.txt
