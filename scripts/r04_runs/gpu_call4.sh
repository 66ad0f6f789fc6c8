#!/bin/bash
# Round-4 call 4: the whole GPU suite on the candidate final build
set -u
OUT=gpurun_out/r04_call4; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.txt 2>&1; tail -30 $OUT/pytest_gpu.txt
cp gpurun_out/parity_r04.jsonl $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
