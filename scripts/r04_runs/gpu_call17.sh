#!/bin/bash
# call 17: timing of the woven attention kernel with DMA / barriers / LDS reads / softmax switched off in turn
set -u
O=gpurun_out/r04_call17; mkdir -p $O
timeout 300 python scripts/r04_runs/diag_attention_woven.py 2>&1 | grep -v amdgpu.ids | tee $O/attention_woven_diag.txt
