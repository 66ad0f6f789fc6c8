#!/bin/bash
# Round-4 call 6: whole GPU suite on the final build + the PMC / trace passes of gpu_final.sh
set -u
OUT=gpurun_out/r04_call6; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
cp gpurun_out/parity_r04.jsonl $OUT/ 2>/dev/null
bash scripts/r04_runs/gpu_final.sh pmc
