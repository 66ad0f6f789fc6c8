#!/bin/bash
# call 22: woven attention kernel, LDS read look-ahead 2 / 3 / 4 / 6 slots
set -u
O=gpurun_out/r04_call22; mkdir -p $O
timeout 300 python scripts/r04_runs/diag_attention_woven.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_woven_lookahead.txt
