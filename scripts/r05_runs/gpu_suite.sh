#!/bin/bash
# Round-5: whole GPU suite + smoke + the default bench command on the current build
set -u
OUT=gpurun_out/r05_suite; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
git rev-parse HEAD > $OUT/HEAD.txt 2>/dev/null
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > $OUT/pytest_gpu.txt 2>&1; tail -22 $OUT/pytest_gpu.txt | cut -c1-300
cp gpurun_out/parity_r05.jsonl $OUT/parity_r05.jsonl 2>/dev/null
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench (default)"; timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-3000
